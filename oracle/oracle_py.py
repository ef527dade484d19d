"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The product package
(vision_amd/) never does: its render path fails loudly when the HIP library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_DIR)
_SO = {3: os.path.join(_DIR, "_build", "liboracle.so"), 4: os.path.join(_DIR, "_build", "liboracle4.so")}
_LIB = {}


def build(force=False):
    """Two builds of the same source: SampledSpectrum of 3 values (spectrum/srgb, spectrum/hero dimension 3) and of 4
    (spectrum/hero with "dimension": 4) — omath.h ORC_SPEC_DIM."""
    src = [os.path.join(_DIR, "oracle.cpp"), os.path.join(_DIR, "omath.h"), os.path.join(_ROOT, "include", "vmk.h")]
    for dim, so in _SO.items():
        if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in src):
            continue
        os.makedirs(os.path.dirname(so), exist_ok=True)
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", f"-DORC_SPEC_DIM={dim}", "-o", so, src[0], "-lpthread"]
        subprocess.check_call(cmd)
    return _SO[3]


def lib(dim=3):
    if dim not in _LIB:
        if not os.path.exists(_SO[dim]):
            build()
        L = C.CDLL(_SO[dim])
        L.orc_spec_dim.restype = C.c_uint32
        assert L.orc_spec_dim() == dim
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                 C.c_void_p]
        L.orc_reset_counters.argtypes = [C.c_void_p]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_test_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_uint32]
        L.orc_integrate_albedo.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.orc_render_aov.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dump_rays.restype = C.c_uint32
        L.orc_dump_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        _LIB[dim] = L
    return _LIB[dim]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """CPU oracle bound to host tables (a vision_amd.host.HostScene keeps them alive)."""

    def __init__(self, host_scene):
        self.host_scene = host_scene
        sc = host_scene.scene
        self.dim = 4 if (sc.spectrum == 1 and sc.spectrum_dimension == 4) else 3
        self._L = lib(self.dim)
        self._h = self._L.orc_scene_create(C.cast(host_scene.tables, C.c_void_p))
        assert self._h, "oracle refused the scene"

    def render(self, params, frame_begin, frame_count, accum=None, tiles=None, threads=0):
        from vision_amd import _abi
        w, h = params.width, params.height
        if accum is None:
            accum = np.zeros((h, w, 4), dtype=np.float32)
        cnt = _abi.Counters()
        self._L.orc_reset_counters(self._h)
        self._L.orc_render(self._h, C.byref(params), frame_begin, frame_count, C.byref(tiles) if tiles else None,
                         _ptr(accum), threads, C.byref(cnt))
        return accum, cnt.as_dict()

    def trace(self, org, dirs, tmax, any_hit=False):
        n = org.shape[0]
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros((n, 4), np.uint32)
        self._L.orc_trace_rays(self._h, n, _ptr(org), _ptr(dirs), _ptr(tmax), int(any_hit), _ptr(out))
        return out

    def render_aov(self, params, frame=0):
        """CPU restatement of the reference's G-buffer kernel (frame_buffer.cpp:156-219): normal / albedo / emission / depth."""
        h, w = params.height, params.width
        c2w = np.array(params.c2w, np.float64).reshape(4, 4).T  # column-major storage
        w2c = np.ascontiguousarray(np.linalg.inv(c2w).T.astype(np.float32)).reshape(-1)  # back to column-major floats
        out = {k: np.zeros((h, w, 4), np.float32) for k in ("normal", "albedo", "emission")}
        r2s = np.array(params.raster_to_sensor, np.float64).reshape(4, 4).T
        s2r = np.ascontiguousarray(np.linalg.inv(r2s).T.astype(np.float32)).reshape(-1)
        out["depth"] = np.zeros((h, w), np.float32)
        out["motion"] = np.zeros((h, w, 2), np.float32)
        self._L.orc_render_aov(self._h, C.byref(params), _ptr(w2c), _ptr(s2r), frame, _ptr(out["normal"]), _ptr(out["albedo"]), _ptr(out["emission"]),
                             _ptr(out["depth"]), _ptr(out["motion"]))
        return out

    def dump_rays(self, params, frame=0, stride=1):
        """Every ray Li() traces for frame `frame` of each stride-th pixel: dict of SoA arrays (org, dir, tmax, kind,
        path, seq) in path order — the input of the traversal-only replay."""
        n = self._L.orc_dump_rays(self._h, C.byref(params), frame, stride, None, 0)
        buf = np.zeros((n, 10), np.uint32)
        self._L.orc_dump_rays(self._h, C.byref(params), frame, stride, _ptr(buf), n)
        f = buf.view(np.float32)
        return {"org": f[:, 0:3].copy(), "dir": f[:, 3:6].copy(), "tmax": f[:, 6].copy(), "kind": buf[:, 7].copy(),
                "path": buf[:, 8].copy(), "seq": buf[:, 9].copy()}

    def test_eval(self, params, kind, inp, out_stride):
        inp = np.ascontiguousarray(inp, np.float32)
        out = np.zeros((inp.shape[0], out_stride), np.float32)
        rc = self._L.orc_test_eval(self._h, C.byref(params) if params is not None else None, kind, inp.shape[0], _ptr(inp),
                                 inp.shape[1], _ptr(out), out_stride)
        assert rc == 0
        return out

    def close(self):
        if self._h:
            self._L.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def test_eval_noscene(kind, inp, out_stride, params=None):
    inp = np.ascontiguousarray(inp, np.float32)
    out = np.zeros((inp.shape[0], out_stride), np.float32)
    rc = lib().orc_test_eval(None, C.byref(params) if params is not None else None, kind, inp.shape[0], _ptr(inp),
                             inp.shape[1], _ptr(out), out_stride)
    assert rc == 0
    return out


def tonemap(params, accum, final_picture=False):
    out = np.zeros_like(accum)
    lib().orc_tonemap(C.byref(params), _ptr(np.ascontiguousarray(accum, np.float32)), int(final_picture), _ptr(out))
    return out


def integrate_albedo(which, res, x, y, z, samples):
    out = np.zeros(2, np.float32)
    lib().orc_integrate_albedo(which, res, x, y, z, samples, _ptr(out))
    return out

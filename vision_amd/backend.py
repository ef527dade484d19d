"""Python binding of libvmk.so (include/vmk.h): the gfx950 megakernel backend.

There is NO CPU fallback: creating a Backend without the HIP library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class BackendError(RuntimeError):
    pass


def lib_path():
    # VMK_LIB: development override used by the tuning scripts to A/B differently compiled builds of the same source
    return os.environ.get("VMK_LIB") or os.path.join(_PKG, "lib", "libvmk.so")


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise BackendError(f"{path} missing — the HIP extension is not built (run __graft_entry__.build()); "
                               "vision_amd has no CPU fallback")
        L = C.CDLL(path)
        L.vmk_last_error.restype = C.c_char_p
        L.vmk_last_error.argtypes = [C.c_void_p]
        L.vmk_abi_version.restype = C.c_uint32
        L.vmk_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.vmk_destroy.argtypes = [C.c_void_p]
        L.vmk_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_build_accel.argtypes = [C.c_void_p]
        L.vmk_set_render_params.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_set_framebuffer.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_reset_accum.argtypes = [C.c_void_p]
        L.vmk_render_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]
        L.vmk_synchronize.argtypes = [C.c_void_p]
        L.vmk_download_accum.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_tonemap.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vmk_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_reset_counters.argtypes = [C.c_void_p]
        L.vmk_set_traversal_counters.argtypes = [C.c_void_p, C.c_int]
        L.vmk_tile_skew.argtypes = [C.c_uint32]
        L.vmk_tile_skew.restype = C.c_uint32
        L.vmk_comm_unique_id.argtypes = [C.c_void_p]
        L.vmk_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.vmk_comm_adopt.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_allreduce_framebuffer.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_allgather_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vmk_comm_synchronize.argtypes = [C.c_void_p]
        L.vmk_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        L.vmk_collect_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.vmk_stream.restype = C.c_void_p
        L.vmk_stream.argtypes = [C.c_void_p]
        L.vmk_accel_info_get.argtypes = [C.c_void_p, C.c_void_p]
        L.vmk_trace_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.POINTER(C.c_float), C.c_uint32]
        L.vmk_self_check.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vmk_render_aov.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vmk_precompute_albedo.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.vmk_test_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        if L.vmk_abi_version() != _abi.ABI_VERSION:
            raise BackendError("libvmk.so ABI version mismatch")
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Backend:
    """One ctx per GPU (not thread-safe), mirroring the call order of Vision's Pipeline::prepare / render."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.vmk_create(device, C.byref(h))
        if rc != 0:
            raise BackendError(self._L.vmk_last_error(None).decode() or f"vmk_create failed ({rc})")
        self._h = h
        self.params = None

    def _check(self, rc):
        if rc != 0:
            raise BackendError(f"{self._L.vmk_last_error(self._h).decode()} (status {rc})")

    def upload_scene(self, host_scene):
        self._check(self._L.vmk_upload_scene(self._h, C.cast(host_scene.tables, C.c_void_p)))
        self._hero = host_scene.scene.spectrum == _abi.SPECTRUM_HERO

    def build_accel(self):
        self._check(self._L.vmk_build_accel(self._h))
        info = _abi.AccelInfo()
        self._check(self._L.vmk_accel_info_get(self._h, C.byref(info)))
        return {n: getattr(info, n) for n, _ in info._fields_}

    def set_render_params(self, params):
        self.params = params
        self._check(self._L.vmk_set_render_params(self._h, C.byref(params)))

    def set_framebuffer(self, device_ptr):
        self._check(self._L.vmk_set_framebuffer(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def reset_accum(self):
        self._check(self._L.vmk_reset_accum(self._h))

    def render_batch(self, frame_begin, frame_count, tiles=None, timed=False):
        ms = C.c_float(0.0)
        self._check(self._L.vmk_render_batch(self._h, frame_begin, frame_count, C.byref(tiles) if tiles else None,
                                             C.byref(ms) if timed else None))
        return ms.value if timed else None

    def synchronize(self):
        self._check(self._L.vmk_synchronize(self._h))

    def enable_kernel_timing(self, enabled=True):
        self._check(self._L.vmk_enable_kernel_timing(self._h, int(bool(enabled))))

    def collect_kernel_ms(self, max_count=4096):
        """HIP-event times (ms) of the asynchronous vmk_render_batch calls since the last collect; waits for the ctx stream."""
        out = np.zeros(max_count, np.float32)
        n = C.c_uint32(0)
        self._check(self._L.vmk_collect_kernel_ms(self._h, _ptr(out), max_count, C.byref(n)))
        return out[:n.value].tolist()

    def download_accum(self):
        out = np.zeros((self.params.height, self.params.width, 4), np.float32)
        self._check(self._L.vmk_download_accum(self._h, _ptr(out)))
        return out

    def tonemap(self, final_picture=False):
        out = np.zeros((self.params.height, self.params.width, 4), np.float32)
        self._check(self._L.vmk_tonemap(self._h, int(final_picture), _ptr(out)))
        return out

    def counters(self):
        c = _abi.Counters()
        self._check(self._L.vmk_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    @property
    def is_hero(self):
        """True when the uploaded scene uses the hero-wavelength spectrum (the vmk_hero.hip megakernel instance)."""
        return bool(getattr(self, "_hero", False))

    def reset_counters(self):
        self._check(self._L.vmk_reset_counters(self._h))

    def set_traversal_counters(self, enabled):
        """Select the megakernel instance with (default) or without the node / triangle tallies in its traversal loops."""
        self._check(self._L.vmk_set_traversal_counters(self._h, int(bool(enabled))))

    # ---- multi-GPU exchange (include/vmk.h: vmk_comm_*) ----
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId through the C-ABI: 128 bytes rank 0 hands to the other ranks (any transport)."""
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        rc = lib().vmk_comm_unique_id(buf)
        if rc != 0:
            raise BackendError(f"{lib().vmk_last_error(None).decode()} (status {rc})")
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._check(self._L.vmk_comm_init(self._h, C.c_char_p(unique_id), rank, world))

    def allreduce_framebuffer(self, recv_device_ptr):
        self._check(self._L.vmk_allreduce_framebuffer(self._h, C.c_void_p(recv_device_ptr)))

    def allgather_framebuffer(self, tiles, recv_device_ptr):
        self._check(self._L.vmk_allgather_framebuffer(self._h, C.byref(tiles), C.c_void_p(recv_device_ptr)))

    def comm_synchronize(self):
        self._check(self._L.vmk_comm_synchronize(self._h))

    @property
    def stream(self):
        return self._L.vmk_stream(self._h)

    def trace(self, org, dirs, tmax, any_hit=False, repeats=1):
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = org.shape[0]
        out = np.zeros((n, 4), np.uint32)
        ms = C.c_float(0.0)
        self._check(self._L.vmk_trace_rays(self._h, n, _ptr(org), _ptr(dirs), _ptr(tmax), int(any_hit), _ptr(out),
                                           C.byref(ms), repeats))
        return out, ms.value

    def test_eval(self, kind, inp, out_stride):
        inp = np.ascontiguousarray(inp, np.float32)
        out = np.zeros((inp.shape[0], out_stride), np.float32)
        self._check(self._L.vmk_test_eval(self._h, kind, inp.shape[0], _ptr(inp), inp.shape[1], _ptr(out), out_stride))
        return out

    def set_auto_self_check(self, enabled=True):
        """vmk_set_auto_self_check: the self check the first vmk_render_batch after a build / parameter change runs by itself."""
        self._L.vmk_set_auto_self_check.argtypes = [C.c_void_p, C.c_int]
        self._check(self._L.vmk_set_auto_self_check(self._h, int(bool(enabled))))

    def self_check(self, max_pixels=0):
        """vmk_self_check: megakernel variant vs unit kernel on frame 0 of a pixel subset; raises BackendError on a mismatch."""
        n, bad = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.vmk_self_check(self._h, max_pixels, C.byref(n), C.byref(bad)))
        return n.value

    def render_aov(self, frame=0):
        """Primary-hit AOV planes of `frame` (include/vmk.h: vmk_render_aov): dict normal / albedo / emission [H, W, 4], depth [H, W], motion [H, W, 2]."""
        h, w = self.params.height, self.params.width
        out = {k: np.zeros((h, w, 4), np.float32) for k in ("normal", "albedo", "emission")}
        out["depth"] = np.zeros((h, w), np.float32)
        out["motion"] = np.zeros((h, w, 2), np.float32)
        self._check(self._L.vmk_render_aov(self._h, frame, _ptr(out["normal"]), _ptr(out["albedo"]), _ptr(out["emission"]), _ptr(out["depth"]),
                                           _ptr(out["motion"])))
        return out

    def precompute_albedo(self, which, res=32, samples=1 << 17):
        """Albedo table `which` (include/vmk.h: vmk_precompute_albedo) as a flat float32 array."""
        n = res * res * (1 if which == 0 else res) * (2 if which in (1, 2) else 1)
        out = np.zeros(n, np.float32)
        self._check(self._L.vmk_precompute_albedo(self._h, which, res, samples, _ptr(out)))
        return out

    def capture_rays(self, pixels_xy, frame=0):
        """Rays the megakernel traces for `frame` of the given pixels (k_test kind 7), as SoA arrays ordered by
        (vertex, ray kind, pixel) — i.e. what the lanes of one wave trace together.  Feeds trace() replays."""
        pix = np.ascontiguousarray(pixels_xy, np.uint32)
        inp = np.concatenate([pix, np.full((pix.shape[0], 1), frame, np.uint32)], axis=1).view(np.float32)
        rec = self.test_eval(7, inp, 1 + 16 * 24)
        nv = rec[:, 0].astype(np.int64)
        v = rec[:, 1:].reshape(-1, 24, 2, 8)          # [path, vertex, closest/shadow, (o, d, tmax, valid)]
        v = np.transpose(v, (1, 2, 0, 3))             # [vertex, kind, path, 8]
        valid = (v[..., 7] == 1.0) & (np.arange(24)[:, None, None] < nv[None, None, :])
        kind = np.broadcast_to(np.arange(2, dtype=np.uint32)[None, :, None], valid.shape)
        r = v[valid]
        return {"org": r[:, 0:3].copy(), "dir": r[:, 3:6].copy(), "tmax": r[:, 6].copy(), "kind": kind[valid].copy()}

    def close(self):
        if getattr(self, "_h", None):
            self._L.vmk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def precompute_albedo_tables(samples=1 << 17, res=32, device=0):
    """The five albedo-compensation tables in vmk_luts order, integrated on the GPU (tools/make_luts.py)."""
    be = Backend(device)
    try:
        return [be.precompute_albedo(w, res, samples) for w in range(5)]
    finally:
        be.close()

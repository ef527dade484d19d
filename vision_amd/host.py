"""Python binding of the C++ host (libvmk_host.so): Vision JSON scene -> flat vmk tables.

The host keeps Vision's plugin namespace (category/type) and JSON schema; see include/vmk_host.h.  Image files (8-bit PNG,
baseline JPEG, Radiance .hdr) are decoded natively by the host (csrc/host/image_codec.h; Vision decodes through ocarina's
Image::load, src/base/mgr/image_pool.cpp:23-28).  Only when the host reports a container it does not decode (progressive JPEG,
16-bit PNG, ...) does this binding fall back to Pillow and hand the pixels over with vmk_host_register_image, as a C++ host that
already holds decoded images would.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None
DEFAULT_LUT_PATH = os.path.join(_PKG, "data", "luts.bin")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_PKG, "lib", "libvmk_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.vmk_host_last_error.restype = C.c_char_p
        L.vmk_host_scene_tables.restype = C.POINTER(_abi.Scene)
        L.vmk_host_render_params.restype = C.POINTER(_abi.RenderParams)
        L.vmk_host_output_fn.restype = C.c_char_p
        L.vmk_host_describe.restype = C.c_char_p
        L.vmk_host_output_spp.restype = C.c_uint32
        L.vmk_host_load_scene.argtypes = [C.c_char_p, C.POINTER(_abi.HostOptions), C.POINTER(C.c_void_p)]
        L.vmk_host_free_scene.argtypes = [C.c_void_p]
        for fn in ("vmk_host_scene_tables", "vmk_host_render_params", "vmk_host_output_spp", "vmk_host_output_fn",
                   "vmk_host_describe"):
            getattr(L, fn).argtypes = [C.c_void_p]
        L.vmk_host_register_image.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.vmk_host_list_images.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
        _LIB = L
    return _LIB


class HostError(RuntimeError):
    pass


def _err():
    return lib().vmk_host_last_error().decode()


def register_images_for(json_path):
    """Decode (Pillow) and register every image file the scene references that exists on disk."""
    L = lib()
    buf = C.create_string_buffer(1 << 16)
    n = L.vmk_host_list_images(json_path.encode(), buf, len(buf))
    if n < 0:
        raise HostError(_err())
    paths = [p for p in buf.value.decode().split("\n") if p]
    for p in paths:
        if not os.path.exists(p) or p.lower().endswith(".hdr"):
            continue  # missing (stand-in handled by the host) or natively decoded
        if p.lower().endswith(".exr"):
            continue  # decoded natively by the C++ host (csrc/host/exr.h)
        from PIL import Image
        im = Image.open(p)
        if im.mode not in ("L", "RGB", "RGBA"):
            im = im.convert("RGBA" if "A" in im.mode else "RGB")
        arr = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
        ch = 1 if arr.ndim == 2 else arr.shape[2]
        rc = L.vmk_host_register_image(p.encode(), arr.shape[1], arr.shape[0], ch, 0, arr.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise HostError(_err())
    return paths


def load_image(path):
    """vmk_host_load_image: decode .png / .jpg / .hdr / .exr natively (C++ host); returns an array [H, W, C] (uint8 or float32)."""
    L = lib()
    L.vmk_host_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    L.vmk_host_free_image.argtypes = [C.c_void_p]
    w, h, ch, fl, px = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int(), C.c_void_p()
    if L.vmk_host_load_image(os.fsencode(path), C.byref(w), C.byref(h), C.byref(ch), C.byref(fl), C.byref(px)) != 0:
        raise HostError(_err())
    try:
        n = w.value * h.value * ch.value
        buf = C.string_at(px, n * (4 if fl.value else 1))
    finally:
        L.vmk_host_free_image(px)
    return np.frombuffer(buf, np.float32 if fl.value else np.uint8).reshape(h.value, w.value, ch.value).copy()


def save_image(path, rgba):
    """vmk_host_save_image: write float RGBA [H, W, 4] as .png (8-bit), .exr (float32, ZIP) or .hdr (RGBE) — Image::save_image of the reference."""
    L = lib()
    L.vmk_host_save_image.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    a = np.ascontiguousarray(rgba, np.float32)
    assert a.ndim == 3 and a.shape[2] == 4
    if L.vmk_host_save_image(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p)) != 0:
        raise HostError(_err())
    return path


def final_picture_mode(fn):
    L = lib()
    L.vmk_host_final_picture_mode.argtypes = [C.c_char_p]
    return int(L.vmk_host_final_picture_mode(os.fsencode(fn)))


class HostScene:
    """Owns the host-side tables of one loaded scene (freed on close())."""

    def __init__(self, json_path, width=0, height=0, max_depth=-1, min_depth=-1, procedural_env=True,
                 drop_unsupported_lights=False, lut_path=None, mediums=False, spectrum=None, missing_assets=None, decode="native"):
        L = lib()
        json_path = os.path.abspath(json_path)
        self.json_path = json_path
        self.image_paths = []
        if decode == "pillow":
            self.image_paths = register_images_for(json_path)
        else:
            L.vmk_host_clear_images()
        opt = _abi.HostOptions(width, height, max_depth, min_depth, int(procedural_env), int(drop_unsupported_lights),
                               (lut_path or DEFAULT_LUT_PATH).encode(), int(mediums),
                               {None: 0, "srgb": 1, "hero": 2, "hero4": 3}[spectrum], {None: 0, "fail": 0, "standin": 1}[missing_assets])
        h = C.c_void_p()
        rc = L.vmk_host_load_scene(json_path.encode(), C.byref(opt), C.byref(h))
        if rc != 0 and decode == "native" and "vmk_host_register_image" in _err():  # a container the native decoders do not handle
            self.image_paths = register_images_for(json_path)
            rc = L.vmk_host_load_scene(json_path.encode(), C.byref(opt), C.byref(h))
        if rc != 0:
            raise HostError(_err())
        self._h = h
        self.tables = L.vmk_host_scene_tables(h)           # POINTER(Scene)
        self.params = L.vmk_host_render_params(h).contents  # RenderParams (owned by the host scene; mutable copy below)
        self.output_spp = int(L.vmk_host_output_spp(h))
        self.output_fn = L.vmk_host_output_fn(h).decode()
        self.description = L.vmk_host_describe(h).decode()

    @property
    def scene(self):
        return self.tables.contents

    def params_copy(self):
        p = _abi.RenderParams()
        C.memmove(C.byref(p), C.byref(self.params), C.sizeof(p))
        return p

    def close(self):
        if getattr(self, "_h", None):
            lib().vmk_host_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

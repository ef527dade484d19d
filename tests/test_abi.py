"""C-ABI surface: libraries load, every symbol the headers declare is exported, struct layouts match (no GPU)."""
import ctypes
import os
import re
import subprocess
import tempfile

from conftest import ROOT
from vision_amd import _abi


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vmk_[a-z_0-9]+)\s*\(", text)))


def test_vmk_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "vision_amd", "lib", "libvmk.so"))
    declared = _declared_functions("vmk.h")
    assert sorted(declared) == sorted(_abi.VMK_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    lib.vmk_abi_version.restype = ctypes.c_uint32
    assert lib.vmk_abi_version() == _abi.ABI_VERSION


def test_host_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "vision_amd", "lib", "libvmk_host.so"))
    declared = [f for f in _declared_functions("vmk_host.h") if f.startswith("vmk_host_")]
    assert sorted(declared) == sorted(_abi.HOST_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_c_header(built):
    names = {"vmk_slot": _abi.Slot, "vmk_material": _abi.Material, "vmk_light": _abi.Light, "vmk_tri_pos": _abi.TriPos,
             "vmk_tri_attr": _abi.TriAttr, "vmk_instance": _abi.Instance, "vmk_texture": _abi.Texture,
             "vmk_luts": _abi.Luts, "vmk_scene": _abi.Scene, "vmk_render_params": _abi.RenderParams,
             "vmk_tiles": _abi.Tiles, "vmk_counters": _abi.Counters, "vmk_accel_info": _abi.AccelInfo,
             "vmk_host_options": _abi.HostOptions}
    src = '#include <stdio.h>\n#include "vmk_host.h"\nint main(void){\n' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    sizes = dict(l.split() for l in out if l)
    for n, t in names.items():
        assert int(sizes[n]) == ctypes.sizeof(t), (n, sizes[n], ctypes.sizeof(t))
    assert ctypes.sizeof(_abi.TriPos) == 48 and ctypes.sizeof(_abi.TriAttr) == 64  # S_tri / shading record sizes


def test_product_fails_loudly_without_gpu(built):
    """No CPU fallback: on a machine without a GPU, creating the backend must raise, not silently run elsewhere."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    from vision_amd.backend import Backend, BackendError
    try:
        Backend(0)
    except BackendError as e:
        assert "no HIP device" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("Backend() succeeded without a GPU")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vision_amd/ or tools/ may import, load or link it (only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    for top in ("vision_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".inl", ".sh")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle_py" not in text and "import oracle" not in text and "liboracle" not in text, os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("oracle_py") == 2 and "cpu_baseline" in bench.split("from oracle import oracle_py")[1][:1200]

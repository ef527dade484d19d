"""The hot path driven from plain C through include/*.h only (tests/c/drive_scene.c): compiles and links everywhere (headers are
C-clean, every symbol resolves); on the GPU box it runs and its film equals the oracle's bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c", "drive_scene.c")


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "drive_scene")
    lib = os.path.join(ROOT, "vision_amd", "lib")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", lib, "-lvmk", "-lvmk_host", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_c_driver_compiles_and_links_against_the_headers(built, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    # without a GPU the product fails loudly at vmk_create — after the host has loaded the scene from C
    scene = os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json")
    lut = os.path.join(ROOT, "vision_amd", "data", "luts.bin")
    r = subprocess.run([exe, scene, "32", "32", "2", os.path.join(str(tmp_path), "o.f32"), lut], capture_output=True, text=True)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 2 and "no HIP device" in r.stderr and "36 triangles" in r.stdout


@pytest.mark.gpu
def test_c_driver_renders_cbox_like_the_oracle(built, tmp_path):
    from vision_amd.host import HostScene
    from oracle import oracle_py
    exe = _build(tmp_path)
    scene = os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json")
    out = os.path.join(str(tmp_path), "film.f32")
    r = subprocess.run([exe, scene, "48", "40", "4", out, os.path.join(ROOT, "vision_amd", "data", "luts.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "self check on 1024 pixels ok" in r.stdout or "self check on" in r.stdout
    film = np.fromfile(out, np.float32).reshape(40, 48, 4)
    hs = HostScene(scene, width=48, height=40)
    ref, _ = oracle_py.OracleScene(hs).render(hs.params_copy(), 0, 4)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    # the save path, from C: the PNG is the 8-bit final picture (two tone maps + sRGB), the EXR the float one without the sRGB curve
    from PIL import Image
    from vision_amd.host import load_image
    png = np.asarray(Image.open(out + ".png")).astype(np.float64) / 255.0
    exr = load_image(out + ".exr").astype(np.float64)
    assert png.shape == (40, 48, 3) and exr.shape == (40, 48, 3) and "saved" in r.stdout
    srgb = np.where(exr <= 0.0031308, 12.92 * exr, 1.055 * np.power(np.maximum(exr, 1e-12), 1 / 2.4) - 0.055)
    assert np.abs(srgb - png).max() < 1.5 / 255.0

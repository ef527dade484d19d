#!/usr/bin/env python3
"""Diagnostic: one frame of a scene under a forced spectrum ("hero" / "hero4" / "srgb") through the megakernel, the path unit kernel
(vmk_test_eval kind 6) and the CPU oracle; prints who disagrees with whom.  usage: diag_spectrum.py <scene> <spectrum> [mediums]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (lives under tests/: it uses the CPU oracle as the checker)
sys.path.insert(0, ROOT)
from vision_amd.backend import Backend
from vision_amd.host import HostScene
from oracle import oracle_py
scene, spectrum = sys.argv[1], sys.argv[2]
W = H = 32
hs = HostScene(os.path.join(ROOT, scene), width=W, height=H, spectrum=spectrum, mediums=len(sys.argv) > 3)
p = hs.params_copy()
be = Backend(0)
be.upload_scene(hs); be.build_accel(); be.set_render_params(p)
be.reset_accum(); be.reset_counters()
be.render_batch(0, 1)
img = be.download_accum()
osc = oracle_py.OracleScene(hs)
ref, _ = osc.render(p, 0, 1)
yy, xx = np.mgrid[0:H, 0:W]
pix = np.stack([xx.ravel(), yy.ravel(), np.zeros(W * H)], 1).astype(np.uint32).view(np.float32)
g6 = be.test_eval(6, pix, 67)
o6 = osc.test_eval(p, 6, pix, 67)
eq = lambda a, b: (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
mega_bad = ~eq(img[..., :3], ref[..., :3]).all(-1).ravel()
unit_bad = ~eq(g6[:, 64:67], o6[:, 64:67]).all(1)
rec_bad = ~eq(g6[:, :64], o6[:, :64]).all(1)
oracle_self = ~eq(o6[:, 64:67], ref.reshape(-1, 4)[:, :3]).all(1)
print(scene, spectrum, "| megakernel != oracle:", int(mega_bad.sum()), "| unit L != oracle L:", int(unit_bad.sum()), "| unit records != oracle records:", int(rec_bad.sum()),
      "| oracle path L != oracle image:", int(oracle_self.sum()))
for i in np.nonzero(mega_bad | unit_bad)[0][:5]:
    print(" px", i % W, i // W, "mega", img.reshape(-1, 4)[i, :3], "unit", g6[i, 64:67], "oracle", o6[i, 64:67])
    for v in range(8):
        a, b = g6[i, v * 8:v * 8 + 8], o6[i, v * 8:v * 8 + 8]
        if not a.any() and not b.any(): break
        print("   v%d %s unit hit %s pdfs %s | oracle hit %s pdfs %s" % (v, "==" if eq(a, b).all() else "!=", a[:2].view(np.uint32), a[4:8], b[:2].view(np.uint32), b[4:8]))

#!/usr/bin/env python3
"""Golden from a reference OUTPUT file: Vision ships res/render_scene/glass-of-water/glass-of-water-1024spp.png, its own
1024-spp render of the scene (1280x720, 8-bit).  This script stores 16x16 block means of that picture (sRGB-encoded
values in [0, 1], 45 x 80 x 3 float16 = 21 KB) as tests/golden/glass_of_water_ref_blocks.npy; the GPU test renders the same
scene with this framework and compares the blocks that are not affected by the mesh missing from the checkout
(models/Mesh000.obj, the poured water).  Data only.   python tools/make_golden_refimage.py
"""
import os
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/res/render_scene/glass-of-water/glass-of-water-1024spp.png"
B = 16
img = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64) / 255.0
h, w, _ = img.shape
blocks = img[:h // B * B, :w // B * B].reshape(h // B, B, w // B, B, 3).mean((1, 3))
out = os.path.join(ROOT, "tests", "golden", "glass_of_water_ref_blocks.npy")
np.save(out, blocks.astype(np.float16))
print(out, blocks.shape, blocks.mean())

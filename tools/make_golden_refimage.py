#!/usr/bin/env python3
"""Goldens from reference OUTPUT files (data only): block means of pictures the reference itself rendered and ships.

  res/render_scene/glass-of-water/glass-of-water-1024spp.png  -> tests/golden/glass_of_water_ref_blocks.npy  (16x16 blocks)

sRGB-encoded values in [0, 1] as float16.  The GPU tests render the same scenes with this framework and compare the parts that
do not depend on assets missing from the checkout.   python tools/make_golden_refimage.py
"""
import os
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/res/render_scene"
# (the two classroom PNGs are not usable as pins: see DESIGN.md section 2)
for src, name, B in ((f"{REF}/glass-of-water/glass-of-water-1024spp.png", "glass_of_water_ref_blocks.npy", 16),):
    img = np.asarray(Image.open(src).convert("RGB")).astype(np.float64) / 255.0
    h, w, _ = img.shape
    blocks = img[:h // B * B, :w // B * B].reshape(h // B, B, w // B, B, 3).mean((1, 3))
    out = os.path.join(ROOT, "tests", "golden", name)
    np.save(out, blocks.astype(np.float16))
    print(out, blocks.shape, blocks.mean())

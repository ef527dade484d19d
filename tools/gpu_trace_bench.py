#!/usr/bin/env python3
"""Traversal-only replay (k_trace): primary camera rays and incoherent interior rays of classroom; Mrays/s + GB/s."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline
scene = os.path.join(ROOT, "scenes/classroom/vision_scene.json")
pipe = Pipeline(scene, width=1920, height=1080); pipe.prepare()
be = pipe.backend
rng = np.random.default_rng(0)
n = 1 << 21
yy, xx = np.mgrid[0:1080, 0:1920]
pix = np.stack([xx.ravel(), yy.ravel(), np.zeros(1920 * 1080)], 1).astype(np.uint32).view(np.float32)[:n]
cam = be.test_eval(5, pix, 6)
s = pipe.host_scene.scene
lo, hi = np.array([-4.4, 0.05, -4.0]), np.array([4.4, 3.0, 6.0])  # the room interior
org = (lo + (hi - lo) * rng.uniform(0, 1, (n, 3))).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
for name, o, dd, anyhit, tm in (("primary closest", cam[:, :3], cam[:, 3:], False, 3e38), ("incoherent closest", org, d, False, 3e38), ("incoherent any-hit", org, d, True, 5.0)):
    be.reset_counters()
    h, ms = be.trace(o, dd, np.full(n, tm, np.float32), any_hit=anyhit, repeats=5)
    c = be.counters()
    rays = c["closest_rays"] + c["shadow_rays"]
    b = c["nodes_visited"] * 128 + c["tris_tested"] * 48 + rays * 44
    print(f"{name}: {ms:.3f} ms/launch, {n / ms / 1e3:.0f} Mrays/s, nodes/ray {c['nodes_visited'] / rays:.1f}, tris/ray {c['tris_tested'] / rays:.1f}, algorithmic {b / 5 / ms / 1e6:.0f} GB/s ({b / 5 / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s), hit frac {(h[:, 0] != 0xFFFFFFFF).mean() if not anyhit else (h[:, 0] == 1).mean():.2f}")

#!/usr/bin/env python3
"""Run-to-run and instance-to-instance determinism of the megakernel at the headline size: the tallying and the
non-tallying instance each render the same 256 frames twice; images and ray counters must agree bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pipe = Pipeline(os.path.join(ROOT, "scenes/classroom/vision_scene.json"), width=1920, height=1080)
pipe.prepare()
be = pipe.backend
res = {}
for name, on in (("count_a", True), ("count_b", True), ("nocount_a", False), ("nocount_b", False)):
    be.set_traversal_counters(on)
    be.reset_accum(); be.reset_counters()
    be.render_batch(0, spp)
    img = be.download_accum()
    c = be.counters()
    res[name] = (img, c)
    print(name, {k: c[k] for k in ("closest_rays", "shadow_rays", "paths", "surface_hits", "tex_fetches")}, flush=True)
base = res["count_a"]
for name in ("count_b", "nocount_a", "nocount_b"):
    img, c = res[name]
    d = (img.view(np.uint32) != base[0].view(np.uint32)).any(-1)
    print(name, "vs count_a: differing pixels", int(d.sum()), "counter deltas", {k: c[k] - base[1][k] for k in ("closest_rays", "shadow_rays", "surface_hits", "tex_fetches")})
    if d.any():
        ys, xs = np.nonzero(d)
        for y, x in list(zip(ys, xs))[:5]:
            print("   px", x, y, img[y, x], base[0][y, x])

#!/usr/bin/env python3
"""Order dependence of the hit selection on bathroom2 at 4K (config 5): the whole frame in one launch, twice, and as eight tile
shards; counts the pixels whose bits differ.  usage (GPU box): python tools/gpu_determinism_c5.py [spp]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd import _abi
from vision_amd.pipeline import Pipeline
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pipe = Pipeline(os.path.join(ROOT, "scenes/bathroom2/vision_scene.json"), width=3840, height=2160, missing_assets="standin", max_depth=64)
pipe.prepare()
be = pipe.backend
def full():
    be.reset_accum(); be.reset_counters(); be.render_batch(0, spp)
    return be.download_accum(), be.counters()
a, ca = full(); b, cb = full()
total = np.zeros_like(a)
for rank in range(8):
    be.reset_accum(); be.render_batch(0, spp, tiles=_abi.Tiles(32, rank, 8)); total += be.download_accum()
for name, img in (("second full launch", b), ("sum of 8 shards", total)):
    d = (img.view(np.uint32) != a.view(np.uint32)).any(-1)
    print(name, "vs first full launch: differing pixels", int(d.sum()), "of", d.size, flush=True)
    ys, xs = np.nonzero(d)
    for y, x in list(zip(ys, xs))[:8]:
        print("   px", x, y, img[y, x], a[y, x])
print("counters", {k: (ca[k], cb[k]) for k in ("closest_rays", "shadow_rays", "surface_hits")})

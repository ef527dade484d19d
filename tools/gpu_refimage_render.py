#!/usr/bin/env python3
"""Render the scenes whose pictures the reference ships, at the pictures' size, and keep the LINEAR accumulation buffer (float16
.npz under gpurun_out/refimg/) so that the comparison with the reference PNGs (tools/refimage_compare.py, which needs
/root/reference and therefore runs in the build container) can be tuned without another GPU call.
usage (GPU box): python tools/gpu_refimage_render.py [spp]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = os.path.join(ROOT, "gpurun_out", "refimg"); os.makedirs(out, exist_ok=True)
JOBS = [  # (name, scene, w, h, options)
    ("cbox_prism_hero4", "scenes/cbox/cbox-prism.json", 1024, 1024, {}),
    ("cbox_prism_srgb", "scenes/cbox/cbox-prism.json", 1024, 1024, {"spectrum": "srgb"}),
    ("staircase", "scenes/staircase/vision_scene.json", 720, 1280, {}),
    ("glass_of_water", "scenes/glass-of-water/vision_scene.json", 1280, 720, {}),
    ("classroom", "scenes/classroom/vision_scene.json", 1280, 720, {}),
    ("coffee", "scenes/coffee/vision_scene.json", 800, 1000, {"missing_assets": "standin"}),
    ("staircase_refcam", "scenes/staircase/vision_scene_refcam.json", 720, 1280, {}),
    ("coffee_refcam", "scenes/coffee/vision_scene_refcam.json", 800, 1000, {"missing_assets": "standin"}),  # res/test_case/coffee: the glass carafe (Mesh010.obj) is not in the checkout
]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, scene, w, h, kw in JOBS:
    if only and name not in only:
        continue
    t0 = time.time()
    pipe = Pipeline(os.path.join(ROOT, scene), width=w, height=h, **kw)
    pipe.prepare(self_check=True)
    done, ms = 0, 0.0
    while done < spp:
        n = min(128, spp - done)
        ms += pipe.render(frames=n); done += n
    img = pipe.frame_buffer.download()[..., :3]
    c = pipe.counters()
    np.savez_compressed(os.path.join(out, name + ".npz"), lin=img.astype(np.float16), spp=spp)
    print(f"{name}: {w}x{h} {spp} spp, kernel {ms:.0f} ms, {(c['closest_rays'] + c['shadow_rays']) / ms / 1e3:.0f} Mrays/s, mean {img.mean((0, 1))}, wall {time.time() - t0:.1f} s", flush=True)
    print(pipe.host_scene.description[:600], flush=True)
    pipe.close()

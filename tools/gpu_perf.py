#!/usr/bin/env python3
"""Quick perf probe: classroom (or any scene) at a given size/spp; prints kernel ms, Mrays/s, visits/ray."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline

scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes/classroom/vision_scene.json")
w, h, spp, reps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 2
spectrum = os.environ.get("VMK_SPECTRUM") or None  # "hero" / "srgb" override of the scene's spectrum block
pipe = Pipeline(scene, width=w, height=h, spectrum=spectrum)
t0 = time.time(); pipe.prepare(); print("prepare", time.time() - t0, "s accel", pipe.accel_info)
if os.environ.get("VMK_NO_TRAV_COUNT"):
    pipe.backend.set_traversal_counters(False)  # the megakernel instance without node / triangle tallies
for r in range(reps):
    pipe.invalidate(); pipe.backend.reset_counters()
    ms = pipe.render(frames=spp)
    c = pipe.counters()
    rays = c["closest_rays"] + c["shadow_rays"]
    bytes_trav = c["nodes_visited"] * 128 + c["tris_tested"] * 48
    print(f"rep {r}: {ms:.2f} ms, {rays / ms / 1e3:.1f} Mrays/s, rays/path {rays / c['paths']:.2f}, nodes/ray {c['nodes_visited'] / rays:.1f}, tris/ray {c['tris_tested'] / rays:.1f}, trav GB/s {bytes_trav / ms / 1e6:.1f}, tex/hit {c['tex_fetches'] / max(c['surface_hits'], 1):.2f}")
if len(sys.argv) > 6:
    pipe.save_result(sys.argv[6])

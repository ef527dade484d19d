#!/usr/bin/env python3
"""Traversal-only replay: capture the rays the megakernel itself traces on classroom (k_test kind 7: one
frame of every 16th 32x32 tile, in the lane order of the megakernel's waves) and replay them through k_trace.
Prints one line per ray class with Mrays/s and algorithmic GB/s (128 B per node record, 48 B per triangle record,
44 B per ray in/out) against the 8 TB/s HBM peak.   usage: gpu_replay.py [scene.json] [width height] [tile_step]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline


def tile_pixels(width, height, tile, step):
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    out = []
    for t in range(0, tx * ty, step):
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        yy, xx = np.mgrid[y0:min(y0 + tile, height), x0:min(x0 + tile, width)]
        out.append(np.stack([xx.ravel(), yy.ravel()], 1))
    return np.concatenate(out).astype(np.uint32)


def replay(pipe, step=16, frame=0, repeats=5, tile_x=4):
    be, p = pipe.backend, pipe.params
    pix = tile_pixels(p.width, p.height, 32, step)
    rays = be.capture_rays(pix, frame)
    res = {"paths": int(pix.shape[0]), "rays": int(rays["kind"].shape[0])}
    for name, kind in (("closest", 0), ("shadow", 1)):
        m = rays["kind"] == kind
        if int(m.sum()) == 0:
            continue
        # the capture is tiled `tile_x` times so that one launch fills the chip (>= 3 M rays), order within a copy kept
        o, d, t = (np.tile(rays[k][m], (tile_x,) + (1,) * (rays[k].ndim - 1)) for k in ("org", "dir", "tmax"))
        n = int(t.shape[0])
        be.reset_counters()
        _, ms = be.trace(o, d, t, any_hit=bool(kind), repeats=repeats)
        c = be.counters()
        traced = c["closest_rays"] + c["shadow_rays"]
        b = (c["nodes_visited"] * 128 + c["tris_tested"] * 48 + traced * 44) / repeats
        res[name] = {"rays": n, "ms": ms, "mrays_s": n / ms / 1e3, "nodes_per_ray": c["nodes_visited"] / traced,
                     "tris_per_ray": c["tris_tested"] / traced, "algorithmic_GBs": b / ms / 1e6, "frac_of_8TBs": b / ms / 1e6 / 8000.0}
    return res


if __name__ == "__main__":
    scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes/classroom/vision_scene.json")
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    step = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    pipe = Pipeline(scene, width=w, height=h); pipe.prepare()
    print("accel", pipe.accel_info)
    print(json.dumps(replay(pipe, step)))

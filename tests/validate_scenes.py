#!/usr/bin/env python3
"""One-off validation on scenes that are not committed fixtures (the reference's other shipped scenes, copied under scenes/_extra/,
git-ignored): vmk_self_check, then the GPU renders the whole picture and the CPU oracle a sample of 32x32 tiles — bit for bit.
usage (GPU box): python tests/validate_scenes.py scenes/_extra/*/vision_scene.json"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd import _abi
from vision_amd.backend import Backend
from vision_amd.host import HostScene
from oracle import oracle_py

W, H, SPP = 256, 144, 2
for path in sys.argv[1:]:
    t0 = time.time()
    try:
        hs = HostScene(path, width=W, height=H, missing_assets="standin")
    except Exception as e:
        print(f"{path}: LOAD FAILED {str(e)[:120]}"); continue
    p = hs.params_copy()
    be = Backend(0)
    try:
        be.upload_scene(hs); info = be.build_accel(); be.set_render_params(p)
        checked = be.self_check()
        be.reset_accum(); be.reset_counters()
        be.render_batch(0, SPP)
        img = be.download_accum(); cg = be.counters()
        ref, co = oracle_py.OracleScene(hs).render(p, 0, SPP, tiles=_abi.Tiles(32, 3, 8))
        owned = ref[..., 3] != 0.0
        same = np.array_equal(img[owned].view(np.uint32), ref[owned].view(np.uint32))
        nst = sum(1 for l in hs.description.split("\n") if "stand-in" in l)
        print(f"{path}: tris {hs.scene.n_tris} mats {hs.scene.n_materials} lights {hs.scene.n_lights} tex {hs.scene.n_textures} stand-ins {nst} depth {p.max_depth} | "
              f"self_check {checked} px ok | oracle tiles {int(owned.sum())} px bit-exact: {same} | finite {bool(np.isfinite(img).all())} | rays/path {(cg['closest_rays'] + cg['shadow_rays']) / max(cg['paths'], 1):.2f} | {time.time() - t0:.1f} s", flush=True)
    except Exception as e:
        print(f"{path}: FAILED {str(e)[:200]}", flush=True)
    finally:
        be.close()

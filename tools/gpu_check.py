#!/usr/bin/env python3
"""Ad-hoc GPU bring-up check (not the test suite): device units + render parity vs the oracle on small cases."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.host import HostScene
from vision_amd.backend import Backend
from oracle import oracle_py

def f2u(x): return np.asarray(x, np.uint32).view(np.float32)

def main():
    be = Backend(0)
    rng = np.random.default_rng(1)
    # units without a scene
    inp = np.zeros((1000, 4), np.float32)
    inp[:, 0] = f2u(rng.integers(0, 2048, 1000)); inp[:, 1] = f2u(rng.integers(0, 2048, 1000)); inp[:, 2] = f2u(rng.integers(0, 5000, 1000)); inp[:, 3] = f2u(rng.integers(0, 2, 1000))
    for kind, arr, ostr in ((0, inp, 8), (1, rng.uniform(-7, 7, (4000, 2)).astype(np.float32), 6), (2, rng.uniform(0, 1, (4000, 2)).astype(np.float32), 8)):
        g = be.test_eval(kind, arr, ostr); o = oracle_py.test_eval_noscene(kind, arr, ostr)
        print(f"unit kind {kind}: bit-exact = {np.array_equal(g.view(np.uint32), o.view(np.uint32))}, max abs diff {np.nanmax(np.abs(g - o))}")
    a3 = np.concatenate([rng.normal(size=(4000, 3)), rng.uniform(0, 1, (4000, 2)), rng.uniform(0.001, 1, (4000, 2)), rng.uniform(1.01, 3, (4000, 1))], axis=1).astype(np.float32)
    g = be.test_eval(3, a3, 8); o = oracle_py.test_eval_noscene(3, a3, 8)
    print(f"unit kind 3: bit-exact = {np.array_equal(g.view(np.uint32), o.view(np.uint32))}, max abs diff {np.nanmax(np.abs(g - o))}")

    for scene_path, res, spp in ((os.path.join(ROOT, "scenes/cbox/cbox_matte.json"), 64, 4), (os.path.join(ROOT, "scenes/cbox/cbox_materials.json"), 64, 4)):
        hs = HostScene(scene_path, width=res, height=res)
        p = hs.params_copy()
        be.upload_scene(hs)
        info = be.build_accel()
        print(scene_path, "accel:", info)
        be.set_render_params(p)
        osc = oracle_py.OracleScene(hs)
        # camera rays
        pix = np.zeros((256, 3), np.float32)
        pix[:, 0] = f2u(rng.integers(0, res, 256)); pix[:, 1] = f2u(rng.integers(0, res, 256)); pix[:, 2] = f2u(rng.integers(0, 64, 256))
        g = be.test_eval(5, pix, 6); o = osc.test_eval(p, 5, pix, 6)
        print("camera rays bit-exact:", np.array_equal(g.view(np.uint32), o.view(np.uint32)), np.abs(g - o).max())
        # traversal parity on camera rays + random rays
        org = np.concatenate([o[:, :3], rng.uniform(-0.9, 0.9, (2000, 3)) + np.array([0, 1, 0])]).astype(np.float32)
        dirs = np.concatenate([o[:, 3:], rng.normal(size=(2000, 3))]).astype(np.float32)
        tmax = np.full(org.shape[0], 3.0e38, np.float32)
        hg, ms = be.trace(org, dirs, tmax); ho = osc.trace(org, dirs, tmax)
        print("closest-hit parity:", np.array_equal(hg, ho), "mismatch", int((hg != ho).any(axis=1).sum()), "of", len(hg), "ms", ms)
        tm2 = rng.uniform(0.1, 2.0, org.shape[0]).astype(np.float32)
        hg, ms = be.trace(org, dirs, tm2, any_hit=True); ho = osc.trace(org, dirs, tm2, any_hit=True)
        print("any-hit parity:", np.array_equal(hg[:, 0], ho[:, 0]), "mismatch", int((hg[:, 0] != ho[:, 0]).sum()))
        # bsdf units per material
        nm = hs.scene.n_materials
        n = 300 * nm
        a4 = np.zeros((n, 12), np.float32)
        a4[:, 0] = f2u(np.repeat(np.arange(nm), 300)); a4[:, 1] = f2u(rng.integers(0, 64, n)); a4[:, 2] = f2u(rng.integers(0, 64, n)); a4[:, 3] = f2u(rng.integers(0, 64, n))
        a4[:, 4:7] = rng.normal(size=(n, 3)); a4[:, 7:10] = rng.normal(size=(n, 3)); a4[:, 10:12] = rng.uniform(0, 1, (n, 2))
        g = be.test_eval(4, a4, 13); o = osc.test_eval(p, 4, a4, 13)
        eq = (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))
        print("bsdf units bit-exact:", bool(eq.all()), "bad rows", int((~eq).any(axis=1).sum()), "of", n, "per-material bad:", [(int(m), int((~eq[a4[:, 0].view(np.uint32) == m]).any(axis=1).sum())) for m in range(nm)])
        # whole-path records
        pix6 = np.zeros((res * res, 3), np.float32)
        yy, xx = np.mgrid[0:res, 0:res]
        pix6[:, 0] = f2u(xx.ravel()); pix6[:, 1] = f2u(yy.ravel()); pix6[:, 2] = f2u(np.zeros(res * res, np.uint32))
        g = be.test_eval(6, pix6, 67); o = osc.test_eval(p, 6, pix6, 67)
        eq = (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))
        bad = np.where(~eq.all(axis=1))[0]
        print("path records bit-exact:", len(bad) == 0, "bad paths", len(bad), "of", res * res)
        for b in bad[:3]:
            col = np.where(~eq[b])[0]
            print("  pixel", b % res, b // res, "first differing fields", col[:8])
            v = col[0] // 8
            print("   gpu vtx", v, g[b, v*8:(v+1)*8], g[b, v*8:v*8+2].view(np.uint32)); print("   cpu vtx", v, o[b, v*8:(v+1)*8], o[b, v*8:v*8+2].view(np.uint32))
            if v > 0: print("   prev gpu", g[b, (v-1)*8:v*8], g[b, (v-1)*8:(v-1)*8+2].view(np.uint32)); print("   prev cpu", o[b, (v-1)*8:v*8])
        # render parity
        be.reset_accum(); be.reset_counters()
        ms = be.render_batch(0, spp, timed=True)
        img_g = be.download_accum(); cg = be.counters()
        img_o, co = osc.render(p, 0, spp)
        d = img_g[..., :3] - img_o[..., :3]
        rel = np.sqrt((d.astype(np.float64) ** 2).sum() / max((img_o[..., :3].astype(np.float64) ** 2).sum(), 1e-30))
        nbad = int((img_g.view(np.uint32) != img_o.view(np.uint32)).any(axis=2).sum())
        print(f"render {res}x{res}x{spp}: kernel {ms:.2f} ms, rel L2 {rel:.3e}, pixels not bit-exact {nbad}/{res*res}, nan {int(np.isnan(img_g).sum())}")
        print(" gpu counters", cg); print(" cpu counters", {k: co[k] for k in ('closest_rays','shadow_rays','paths','surface_hits')})
        np.save(os.path.join(ROOT, "gpurun_out", f"img_gpu_{os.path.basename(scene_path)}.npy"), img_g)
        np.save(os.path.join(ROOT, "gpurun_out", f"img_cpu_{os.path.basename(scene_path)}.npy"), img_o)

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()

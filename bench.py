#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path on classroom 1920x1080 (BASELINE.json config 3).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one megakernel batch: `--spp-per-step` consecutive 1-spp frames of the whole image (K steps accumulate
K*spp samples per pixel; the default 4 x 256 is the metric's 1024 spp — the largest batch whose staging planes fit one
launch at 1080p; the drain of the persistent kernel at the end of a launch costs ~2.5 ms whatever the batch, so larger
batches lose less to it, most of all when 8 GPUs share the image).  Scene tables and BVH are resident in HBM before
the timed region.  With N ranks the image tiles are sharded (weak per-pixel cost, strong over the image: the total
work is fixed, so "scaling" is "strong") and every step ends with ONE all-reduce of the float4 framebuffer (RCCL).
value = closest + shadow rays traced by all ranks in the timed steps / max-over-ranks wall time.
Rank 0 prints one JSON line.  cpu_baseline (rank 0, N=1 only) times the CPU oracle (oracle/, test infrastructure) on
a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def algorithmic_bytes(c, pixels, launches, env_lit, texel_bytes):
    """SURVEY.md §8d / DESIGN.md §measurement: touched bytes of the megakernel, from its in-kernel counters.
    traversal: 128 B per BVH4 node record + 48 B per triangle record; per shaded hit: 48 + 64 B triangle records + 52 B
    instance row + >= 32 B material block (lower bound: diffuse); per bilinear texture lookup 4 texels; per NEE sample
    the alias entries (12 B each) and, for the environment light, 4 RGBA32F texels; film: 16 B read + 16 B write."""
    trav = c["nodes_visited"] * 128 + c["tris_tested"] * 48
    hit = c["surface_hits"] * (48 + 64 + 52 + 32)
    tex = c["tex_fetches"] * 4 * texel_bytes
    nee = c["shadow_rays"] * ((24 + 8 + 64) if env_lit else (12 + 48 + 64 + 52))
    film = pixels * 32 * launches
    return trav + hit + tex + nee + film, trav


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "classroom", "vision_scene.json"))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-replay", action="store_true", help="skip the traversal-only replay of the megakernel's own rays")
    ap.add_argument("--no-self-check", action="store_true", help="skip vmk_self_check (profiling runs: keeps every k_render dispatch a timed step)")
    ap.add_argument("--save", default=None, help="write the final tone-mapped picture (rank 0)")
    a = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world != 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")  # RCCL on ROCm

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist:
        dist.barrier()
    from vision_amd.pipeline import Pipeline
    from vision_amd import _abi

    pipe = Pipeline(a.scene, device=local_rank, width=a.width, height=a.height)
    pipe.prepare()
    # the megakernel variant about to be timed agrees bit for bit with the unit kernel (raises otherwise)
    checked = 0 if a.no_self_check else pipe.backend.self_check()
    params = pipe.params
    pixels = params.width * params.height
    dev = torch.device("cuda", local_rank)
    fb = torch.zeros((params.height, params.width, 4), dtype=torch.float32, device=dev)
    pipe.use_torch_framebuffer(fb)
    full = torch.zeros_like(fb) if world > 1 else fb
    pipe.set_tiles(a.tile, rank, world)
    spp = a.spp_per_step

    def sync_all():
        pipe.backend.synchronize()
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    kernel_ms = []

    def step():
        ms = pipe.render(frames=spp, timed=True)  # HIP events on the ctx stream around the megakernel launch
        kernel_ms.append(ms)
        if dist:  # the path's one exchange step: framebuffer all-reduce (disjoint tiles, x + 0 is exact)
            full.copy_(fb)
            dist.all_reduce(full, op=dist.ReduceOp.SUM)

    pipe.invalidate()
    for _ in range(a.warmup):
        step()
    sync_all()
    pipe.backend.reset_counters()
    kernel_ms.clear()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    c = pipe.counters()
    keys = ["closest_rays", "shadow_rays", "nodes_visited", "tris_tested", "paths", "surface_hits", "tex_fetches"]
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([c[k] for k in keys], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        c_all = {k: int(v) for k, v in zip(keys, tot.tolist())}
    else:
        c_all = c
    rays = c_all["closest_rays"] + c_all["shadow_rays"]

    if rank == 0:
        sc = pipe.host_scene.scene
        env_lit = sc.env_light != _abi.INVALID
        # rank-0 launch: its own counters, its own kernel times
        b_all, b_trav = algorithmic_bytes(c, pixels // world, a.steps, env_lit, 4)
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        achieved = b_all / a.steps / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s on classroom@1024spp, 1/2/4/8 GPUs; HBM GB/s vs roofline",
            "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "classroom scene assets (CC-BY, shipped by the reference) + procedural stand-in for the stripped HDRI",
            "config": {"workload": f"classroom {params.width}x{params.height}, max_depth {params.max_depth}, min_depth {params.min_depth}, "
                                   f"{spp} spp per step x {a.steps} steps = {spp * a.steps} spp, env-lit, box filter",
                       "triangles": int(sc.n_tris), "spp_per_step": spp, "tile": a.tile, "parallelism": f"tiles/{world}"},
            "self_check": f"megakernel == unit kernel on {checked} pixels of frame 0 (bit-exact)" if checked else "skipped",
            "rays_per_path": rays / max(c_all["paths"], 1),
            "nodes_per_ray": c_all["nodes_visited"] / max(rays, 1), "tris_per_ray": c_all["tris_tested"] / max(rays, 1),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_render", "avg_kernel_ms": avg_ms,
                         "traversal_GBs": b_trav / a.steps / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0},
        }
        # measured device-to-device copy bandwidth of this GPU in the same run (quote the fraction against both)
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst = torch.empty_like(src)  # 1 GiB each
        dst.copy_(src); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize(dev)
        copy_gbs = 8 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9  # read + write
        del src, dst
        out["roofline"]["measured_copy_GBs"] = copy_gbs
        out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
        # traversal-only replay: the rays the megakernel traces (captured by k_test kind 7, bit-identical to the
        # oracle's) replayed through k_trace; algorithmic bytes = 128 B per node visit + 48 B per triangle test + 44 B per ray
        if world == 1 and not a.no_replay:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from gpu_replay import replay
            rp = replay(pipe)
            out["roofline"]["traversal_replay"] = {k: {"mrays_s": rp[k]["mrays_s"], "achieved_GBs": rp[k]["algorithmic_GBs"], "frac": rp[k]["frac_of_8TBs"],
                                                       "nodes_per_ray": rp[k]["nodes_per_ray"], "tris_per_ray": rp[k]["tris_per_ray"]}
                                                   for k in ("closest", "shadow") if k in rp}
        # HBM traffic of k_render from separate rocprofv3 --pmc passes of this same command (tools/summarize_profiles.py);
        # only attached when the profiled workload matches this run, otherwise null.
        pmc_path = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if world == 1 and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("spp_per_step") == spp and (a.width, a.height) == (1920, 1080):
                out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = f"profiles/{pmc['tag']}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, (2*FETCH+WRITE)*1024 per launch)"
                out["roofline"]["algorithmic_bytes_per_launch"] = b_all / a.steps
        if world == 1 and not a.no_cpu_baseline:
            from oracle import oracle_py
            osc = oracle_py.OracleScene(pipe.host_scene)
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, int(os.environ.get("VMK_CPU_THREADS", "32"))))  # the GPU box grants a CPU share, not the whole host
            sub, nf = 4, 8  # every 4th 32x32 tile of the same 1920x1080 image, frames 0..7 (about 10-20 s on 32 threads)
            t1 = time.perf_counter()
            _, cc = osc.render(params, 0, nf, tiles=_abi.Tiles(a.tile, 0, sub), threads=cores)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": (cc["closest_rays"] + cc["shadow_rays"]) / dt / 1e6, "unit": "Mrays/s", "cores": cores,
                                   "kind": "port", "sample": f"same scene/resolution, every {sub}th {a.tile}x{a.tile} tile, frames 0-{nf - 1} "
                                                             f"({cc['paths']} paths, {dt:.1f} s); build CPU restatement (not Vision/ocarina)"}
        print(json.dumps(out), flush=True)
        if a.save:
            if world > 1:
                pipe.backend.set_framebuffer(full.data_ptr())
            pipe.save_result(a.save)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path (BASELINE.json: classroom 1920x1080 @ 1024 spp).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one megakernel batch: `--spp-per-step` consecutive 1-spp frames of the whole image (K steps accumulate
K*spp samples per pixel; the default 4 x 256 is the metric's 1024 spp — the largest batch whose staging planes fit one
launch at 1080p).  Scene tables and BVH are resident in HBM before the timed region.  With N ranks the image tiles are
sharded (include/vmk.h vmk_tiles; total work fixed -> "scaling": "strong") and every step ends with the path's one
exchange, behind the C-ABI: vmk_allreduce_framebuffer (RCCL, on the ctx's exchange stream; the next step's megakernel
starts at once and only its film resolve waits for it).  torch.distributed is used to start the ranks, to hand rank 0's
RCCL id to the others and for the barrier / max-over-ranks bookkeeping.
value = closest + shadow rays traced by all ranks in the timed steps / max-over-ranks wall time.

The timed steps run the megakernel instance WITHOUT the node / triangle tallies (two adds in the innermost traversal
loops); the tallies behind `roofline.achieved` come from a sibling pass over the same frames afterwards — rays are a pure
function of (pixel, frame), and the sibling's ray counts are asserted equal to the timed steps'.
`--config c4 | c5` benches the other single-/multi-GPU BASELINE configurations (glass-of-water hero depth 64; bathroom2 4K).
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


def cpu_budget():
    """Hardware threads this process may actually run on: the scheduler affinity, cut down to the cgroup's CPU quota when there is one
    (a GPU box hands a job 16 CPUs of a 256-thread host; 256 oracle threads on 16 CPUs time-slice and the measured rate falls)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    if os.environ.get("VMK_CPU_THREADS"):
        n = max(1, min(n, int(os.environ["VMK_CPU_THREADS"])))
    return n


def quick_config(name, device, tile):
    """One BASELINE configuration measured briefly on one GPU, after the headline's timed region: a warm-up step and a timed step of the
    non-tallying megakernel instance, then the same frames through the tallying instance for the algorithmic bytes."""
    from vision_amd.pipeline import Pipeline
    from vision_amd import _abi
    scene_rel, w, h, host_kw, workload = CONFIGS[name]
    spp = 64 if w * h > 4000000 else 256
    pipe = Pipeline(os.path.join(ROOT, scene_rel), device=device, width=w, height=h, **host_kw)
    try:
        pipe.prepare()
        be = pipe.backend
        be.set_traversal_counters(False)
        pipe.render(frames=spp)               # warm-up (and the automatic self check of the first batch)
        be.reset_counters()
        ms = pipe.render(frames=spp)          # timed: HIP events around the launches of this batch
        c = pipe.counters()
        be.set_traversal_counters(True)
        be.reset_counters()
        be.render_batch(spp, spp)
        be.synchronize()
        sib = pipe.counters()
        c = dict(c, nodes_visited=sib["nodes_visited"], tris_tested=sib["tris_tested"])
        rays = c["closest_rays"] + c["shadow_rays"]
        b_all, _ = algorithmic_bytes(c, w * h, 1, pipe.host_scene.scene.env_light != _abi.INVALID, 4)
        gbs = b_all / (ms * 1e-3) / 1e9
        return {"workload": workload, "value": rays / ms / 1e3, "unit": "Mrays/s", "ms_per_step": ms, "spp_per_step": spp,
                "frac": gbs / HBM_PEAK_GBS, "achieved_GBs": gbs, "nodes_per_ray": c["nodes_visited"] / max(rays, 1), "rays_per_path": rays / max(c["paths"], 1)}
    finally:
        pipe.close()


CONFIGS = {  # BASELINE.json configs -> (scene, width, height, host options, description)
    "c3": ("scenes/classroom/vision_scene.json", 1920, 1080, {}, "classroom 1920x1080, env-lit, box filter"),
    "c2": ("scenes/cbox/cbox_matte.json", 1024, 1024, {}, "cbox 1024x1024, matte-only BSDFs"),
    "c4": ("scenes/glass-of-water/vision_scene.json", 1024, 1024, {"spectrum": "hero", "max_depth": 64, "min_depth": 3},
           "glass-of-water 1024x1024, spectrum hero (spectral glass), max_depth 64, min_depth 3"),
    "c5": ("scenes/bathroom2/vision_scene.json", 3840, 2160, {"max_depth": 64, "missing_assets": "standin"},
           "bathroom2 3840x2160, max_depth 64, stand-ins for the assets stripped from the reference checkout"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=0, help="default: 256 (64 at 4K, where 256 planes do not fit one launch)")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="BASELINE.json configuration (default c3 = the headline)")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "allgather"], help="form of the per-step exchange (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-replay", action="store_true", help="skip the traversal-only replay of the megakernel's own rays")
    ap.add_argument("--no-self-check", action="store_true", help="skip vmk_self_check (profiling runs: keeps every k_render dispatch a timed step)")
    ap.add_argument("--no-sibling", action="store_true", help="skip the counting sibling pass (roofline.achieved is then null)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short measurements of BASELINE configs c2 / c4 / c5 after the timed region (N = 1, config c3 only)")
    ap.add_argument("--force-exchange", action="store_true", help="rehearsal on one GPU: run the per-step C-ABI exchange with a 1-rank communicator")
    ap.add_argument("--rehearse-dist", action="store_true", help="rehearsal on one GPU under torch.distributed.run --nproc-per-node 1: initialise the RCCL process "
                    "group although WORLD_SIZE is 1, so that every `if dist:` branch of the N > 1 path runs (with --force-exchange also the C-ABI communicator beside torch's)")
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
    if world > 1 or a.rehearse_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")  # RCCL on ROCm: rendezvous, barrier and bookkeeping only — the data path's collective is vmk_allreduce_framebuffer

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist:
        dist.barrier()
    from vision_amd.pipeline import Pipeline
    from vision_amd.backend import Backend
    from vision_amd import _abi

    scene_rel, cw, ch, host_kw, workload = CONFIGS[a.config]
    scene_path = a.scene or os.path.join(ROOT, scene_rel)
    width, height = a.width or cw, a.height or ch
    spp = a.spp_per_step or (64 if width * height > 4000000 else 256)
    pipe = Pipeline(scene_path, device=local_rank, width=width, height=height, **host_kw)
    pipe.prepare()
    be = pipe.backend
    # the megakernel variant about to be timed agrees bit for bit with the unit kernel (raises otherwise)
    if a.no_self_check:
        be.set_auto_self_check(False)  # (profiling runs: the self check's own 1-spp launch would count as a k_render dispatch)
    checked = 0 if a.no_self_check else be.self_check()
    params = pipe.params
    pixels = params.width * params.height
    dev = torch.device("cuda", local_rank)
    fb = torch.zeros((params.height, params.width, 4), dtype=torch.float32, device=dev)
    pipe.use_torch_framebuffer(fb)
    exchange = world > 1 or a.force_exchange
    full = torch.zeros_like(fb) if exchange else fb
    pipe.set_tiles(a.tile, rank, world)
    exchange_via = "none"
    if exchange:  # RCCL communicator behind the C-ABI: rank 0's id travels over the rendezvous backend
        try:
            idt = torch.zeros(_abi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(Backend.comm_unique_id()), dtype=torch.uint8))
            if dist:
                dist.broadcast(idt, 0)
            be.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
            exchange_via = a.exchange + " behind the C-ABI (vmk_" + a.exchange + "_framebuffer over RCCL), overlapped with the next step"
        except Exception as e:  # never lose the multi-GPU measurement to the communicator bootstrap: same collective through torch
            exchange_via = f"torch.distributed all_reduce (C-ABI communicator unavailable: {str(e)[:160]})"
        if dist:  # every rank must take the same path
            ok = torch.tensor([0 if exchange_via.startswith("torch") else 1], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and not exchange_via.startswith("torch"):
                exchange_via = "torch.distributed all_reduce (another rank could not create the C-ABI communicator)"
    tiles_for_exchange = pipe.tiles if pipe.tiles is not None else _abi.Tiles(a.tile, 0, 1)

    def sync_all():
        be.synchronize()
        be.comm_synchronize()
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def step():
        pipe.render(frames=spp, timed=False)  # asynchronous; HIP events on the ctx stream bracket the launch (collect_kernel_ms)
        if exchange:  # the path's one exchange step (disjoint tiles, x + 0 is exact), overlapped with the next step's megakernel
            if exchange_via.startswith("torch"):
                be.synchronize()
                full.copy_(fb)
                if dist:
                    dist.all_reduce(full, op=dist.ReduceOp.SUM)
                torch.cuda.current_stream().synchronize()  # the copy must have read fb before the next film resolve writes it
            elif a.exchange == "allreduce":
                be.allreduce_framebuffer(full.data_ptr())
            else:
                be.allgather_framebuffer(tiles_for_exchange, full.data_ptr())

    be.enable_kernel_timing(True)
    be.set_traversal_counters(False)  # timed instance: no tallies in the traversal loops (sibling pass below supplies them)
    pipe.invalidate()
    for _ in range(a.warmup):
        step()
    sync_all()
    be.reset_counters()
    be.collect_kernel_ms()
    first_frame = pipe.frame_index()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = be.collect_kernel_ms()
    c = pipe.counters()
    if a.save and rank == 0:
        if exchange:
            be.set_framebuffer(full.data_ptr())
        pipe.save_result(a.save)
        if exchange:
            be.set_framebuffer(fb.data_ptr())
    # ---- sibling pass: the same frames through the tallying instance (film contents are not used afterwards) ----
    sib = None
    if not a.no_sibling:
        be.set_traversal_counters(True)
        be.reset_counters()
        for k in range(a.steps):
            be.render_batch(first_frame + k * spp, spp, tiles=pipe.tiles)
        be.synchronize()
        sib = pipe.counters()
        # Same frames, same rays — up to the order-dependent near-ties of the traversal: a candidate hit whose computed t falls outside
        # its own box by more than the cull margin is found or not depending on when the quad's culling bound tightened, which
        # depends on the wave's phase scheduling.  Measured: ~2 paths in 1e9; anything beyond 1e-7 is a bug.
        assert sib["paths"] == c["paths"]
        sib_delta = {k: sib[k] - c[k] for k in ("closest_rays", "shadow_rays", "surface_hits", "tex_fetches")}
        for k, d in sib_delta.items():
            assert abs(d) <= 1e-7 * max(c[k], 1) + 2, f"sibling pass traced different work ({k}: {sib[k]} vs {c[k]})"
        c = dict(c, nodes_visited=sib["nodes_visited"], tris_tested=sib["tris_tested"])
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
        # rank-0 launches: their own tallies (sibling pass), their own kernel times (HIP events on the ctx stream)
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        b_all = b_trav = achieved = None
        if sib is not None:
            b_all, b_trav = algorithmic_bytes(c, pixels // world, a.steps, env_lit, 4)
            achieved = b_all / a.steps / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s on classroom@1024spp, 1/2/4/8 GPUs; HBM GB/s vs roofline",
            "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "scene assets shipped by the reference (CC-BY / CC0) + declared stand-ins for the stripped ones: " + "; ".join(
                l.split(" ", 1)[1] for l in pipe.host_scene.description.split("\n") if "stand-in" in l)[:400],
            "config": {"workload": f"{workload}: {params.width}x{params.height}, max_depth {params.max_depth}, min_depth {params.min_depth}, "
                                   f"{spp} spp per step x {a.steps} steps = {spp * a.steps} spp",
                       "baseline_config": a.config, "triangles": int(sc.n_tris), "spp_per_step": spp, "tile": a.tile, "parallelism": f"tiles/{world}",
                       "exchange": exchange_via},
            "self_check": f"megakernel == unit kernel on {checked} pixels of frame 0 (bit-exact)" if checked else "skipped",
            "rays_per_path": rays / max(c_all["paths"], 1),
            "nodes_per_ray": c_all["nodes_visited"] / max(rays, 1), "tris_per_ray": c_all["tris_tested"] / max(rays, 1),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved is not None else None,
                         "traffic": None, "kernel": "k_render", "avg_kernel_ms": avg_ms,
                         "traversal_GBs": b_trav / a.steps / (avg_ms * 1e-3) / 1e9 if (b_trav is not None and avg_ms > 0) else None,
                         "achieved_is": "ALGORITHMIC (touched) bytes of SURVEY 8(d) per second — a cache-served rate, see hbm_counter_frac and limiter",
                         "tallies_from": ("sibling pass over the same frames with the tallying kernel instance; ray-count deltas vs the timed steps "
                                          f"(order-dependent near-ties, asserted <= 1e-7): {sib_delta}") if sib is not None else None},
        }
        rl = out["roofline"]
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
        rl["measured_copy_GBs"] = copy_gbs
        rl["frac_of_measured_copy"] = achieved / copy_gbs if achieved is not None else None
        # traversal-only replay: the rays the megakernel traces (captured by k_test kind 7, bit-identical to the
        # oracle's) replayed through k_trace; algorithmic bytes = 128 B per node visit + 48 B per triangle test + 44 B per ray
        if world == 1 and not a.no_replay and not pipe.backend.is_hero:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from gpu_replay import replay
            rp = replay(pipe)
            rl["traversal_replay"] = {k: {"mrays_s": rp[k]["mrays_s"], "achieved_GBs": rp[k]["algorithmic_GBs"], "frac": rp[k]["frac_of_8TBs"],
                                          "nodes_per_ray": rp[k]["nodes_per_ray"], "tris_per_ray": rp[k]["tris_per_ray"]}
                                      for k in ("closest", "shadow") if k in rp}
        # Counter-derived figures of k_render from separate rocprofv3 --pmc passes of this same command (tools/gpu_profile.sh ->
        # tools/summarize_profiles.py -> profiles/latest_pmc.json).  Attached only when that profile was taken from THIS build
        # (device_build_id) on THIS workload; a stale file is refused, not quoted.
        pmc_path = os.path.join(ROOT, "profiles", "latest_pmc.json")
        rl["limiter"] = "unprofiled build: run tools/gpu_profile.sh"
        if world == 1 and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("head") != g.device_build_id():
                rl["pmc_refused"] = f"profiles/latest_pmc.json was taken from build {pmc.get('head')}, this is {g.device_build_id()}"
            elif pmc.get("spp_per_step") != spp or pmc.get("config") != a.config:
                rl["pmc_refused"] = "profiles/latest_pmc.json is for another workload"
            else:
                t_s = avg_ms * 1e-3
                rl["traffic"] = pmc["traffic_bytes_per_launch"]
                rl["traffic_source"] = f"profiles/{pmc['tag']}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, (2*FETCH+WRITE)*1024 per launch)"
                rl["algorithmic_bytes_per_launch"] = b_all / a.steps if b_all is not None else None
                rl["hbm_counter_frac"] = pmc["traffic_bytes_per_launch"] / t_s / 1e9 / HBM_PEAK_GBS
                if "SQ_INSTS_VALU" in pmc:
                    valu = pmc["SQ_INSTS_VALU"]  # wave-instructions per launch
                    rays_launch = rays / a.steps
                    per_simd_clk = 256 * 4 * 2.4e9  # CUs x SIMDs x max clock (MI355X_MICROARCH.md)
                    rl["valu"] = {"wave_instructions_per_launch": valu, "per_ray": valu / rays_launch, "rate_G_per_s": valu / t_s / 1e9,
                                  "ceiling_4cyc_G_per_s": per_simd_clk / 4 / 1e9, "frac_4cyc": valu / t_s / (per_simd_clk / 4),
                                  "ceiling_2cyc_G_per_s": per_simd_clk / 2 / 1e9, "frac_2cyc": valu / t_s / (per_simd_clk / 2),
                                  "valu_busy_pct": pmc.get("VALUBusy"), "lane_utilization_pct": pmc.get("VALUUtilization"),
                                  "note": "measured on this chip at 6 waves per SIMD (tools/experiments/valu_rate.hip, profiles/r03_valu_rate.log): v_fma_f32 2.8-3.1 cycles per wave64 instruction per SIMD, "
                                          "a v_min/v_max/v_add_u32/v_cndmask mix 4.10, quad-permute DPP moves 4.15, v_pk_fma_f32 5.2 - the traversal's mix is the 4-cycle kind; VALUBusy is the hardware's own figure",
                                  "source": f"profiles/{pmc['tag']}_pmc.json"}
                    rl["limiter"] = ("VALU issue: %.2f of the 4-cycle-per-instruction rate this instruction mix gets (measured ceiling, DESIGN.md 4.3); VALUBusy %.0f %%, "
                                     "%.0f %% of the lanes active - about 40 %% of the instructions are BVH node steps at 65 %% quad occupancy, 16 %% leaf phases, 44 %% shading "
                                     "(profiles/r03_occ_probe.log); HBM sees %.2f of its peak, the touched-bytes rate is served by L2 / Infinity Cache"
                                     ) % (rl["valu"]["frac_4cyc"], pmc.get("VALUBusy", float("nan")), pmc.get("VALUUtilization", float("nan")), rl["hbm_counter_frac"])
        if world == 1 and not a.no_cpu_baseline:
            from oracle import oracle_py
            osc = oracle_py.OracleScene(pipe.host_scene)
            cores = cpu_budget()
            sub, nf = 4, 8  # every 4th 32x32 tile of the same image, frames 0..7 (about 10-30 s; the oracle's work queue is tile-granular)
            t1 = time.perf_counter()
            _, cc = osc.render(params, 0, nf, tiles=_abi.Tiles(a.tile, 0, sub), threads=cores)
            dt = time.perf_counter() - t1
            v = (cc["closest_rays"] + cc["shadow_rays"]) / dt / 1e6
            out["cpu_baseline"] = {"value": v, "unit": "Mrays/s", "cores": cores, "per_core": v / cores,
                                   "kind": "port", "sample": f"same scene/resolution, every {sub}th {a.tile}x{a.tile} tile, frames 0-{nf - 1} "
                                                             f"({cc['paths']} paths, {dt:.1f} s) on the {cores} hardware threads this process may use (affinity and cgroup quota); build CPU restatement (not Vision/ocarina)"}
        if world == 1 and a.config == "c3" and not a.no_other_configs and not a.scene:
            # the other BASELINE configurations that fit one GPU, measured after the timed region (each a warm-up + one timed step)
            pipe.close()
            out["other_configs"] = {}
            for name in ("c2", "c4", "c5"):
                try:
                    out["other_configs"][name] = quick_config(name, local_rank, a.tile)
                except Exception as e:  # never lose the headline line to a side measurement
                    out["other_configs"][name] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()


if __name__ == "__main__":
    main()

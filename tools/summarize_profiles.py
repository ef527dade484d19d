#!/usr/bin/env python3
"""Turn a rocprofv3 run of bench.py (gpurun_out/prof/{kt,pmc_fetch,pmc_write,pmc_l2}) into committed summaries under
profiles/: <tag>_kernel_stats.csv (top rows of --kernel-trace --stats), <tag>_pmc.json (per-launch HBM traffic of
k_render, MI355X_MICROARCH.md §HBM corrections) and <tag>_summary.md.   usage: summarize_profiles.py <tag> <spp_per_step> [prof_dir] [config]"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, spp = sys.argv[1], int(sys.argv[2])
prof = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "prof")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

def short(n):
    return n.split("(")[0][-60:]

def newest(pattern):  # gpurun merges every call's files into the same local directory: take the latest run
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []

stats = list(csv.DictReader(open(newest(os.path.join(prof, "kt", "*", "*_kernel_stats.csv"))[0])))
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in stats[:12]:
        f.write(",".join([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]) + "\n")
kr = max((r for r in stats if "k_render" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))

def pmc(name):
    f = newest(os.path.join(prof, name, "*", "*_counter_collection.csv"))
    if not f:
        return {}, {}
    rows = [r for r in csv.DictReader(open(f[0])) if "k_render" in r["Kernel_Name"]]
    agg = collections.defaultdict(float)
    for r in rows:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    n = len({r["Dispatch_Id"] for r in rows})
    meta = {k: rows[0][k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")} if rows else {}
    return {k: v / max(n, 1) for k, v in agg.items()}, meta

import sys as _sys
_sys.path.insert(0, ROOT)
import __graft_entry__ as _g
config = sys.argv[4] if len(sys.argv) > 4 else "c3"
allc, meta = {}, {}
for d in sorted(glob.glob(os.path.join(prof, "pmc*"))):  # every --pmc pass of tools/gpu_profile.sh: per-launch means of k_render
    vals, m = pmc(os.path.basename(d))
    allc.update(vals)
    meta = meta or m
fetch_kb, write_kb = allc.get("FETCH_SIZE", 0.0), allc.get("WRITE_SIZE", 0.0)
# MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
# stream (64 B tallied per 128 B request) -> doubled; WRITE_SIZE is exact.  (16-B gathers are "uncalibrated" per the guide:
# the doubled figure is an upper estimate for the read side.)
traffic = (2.0 * fetch_kb + write_kb) * 1024.0
res = {"tag": tag, "kernel": "k_render", "spp_per_step": spp, "config": config, "head": _g.device_build_id(),
       "avg_kernel_ns": float(kr["AverageNs"]), "calls": int(kr["Calls"]),
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb, "traffic_bytes_per_launch": traffic,
       "traffic_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024.0,
       "L2_hit_rate": allc.get("TCC_HIT_sum", 0) / max(allc.get("TCC_HIT_sum", 0) + allc.get("TCC_MISS_sum", 0), 1), "dispatch": meta}
n_disp = {}
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
          "VALUBusy", "VALUUtilization", "MemUnitBusy", "LDSBankConflict"):
    if k in allc:
        res[k] = allc[k]
l2 = allc
json.dump(res, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
bench_line = ""
for l in open(os.path.join(prof, "bench_kt.log")):
    if l.startswith('{"metric"'):
        bench_line = l.strip()
with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 summary of `python3 bench.py --config {config} --steps 2 --warmup 1 --spp-per-step {spp} --no-cpu-baseline --no-replay --no-self-check --no-sibling` on MI355X (tools/gpu_profile.sh)\n\n")
    f.write(f"* `--kernel-trace --stats`: k_render {kr['Calls']} calls, average {float(kr['AverageNs'])/1e6:.3f} ms (min {float(kr['MinNs'])/1e6:.3f}, max {float(kr['MaxNs'])/1e6:.3f}), {kr['Percentage']} % of GPU time — see `{tag}_kernel_stats.csv`.\n")
    f.write(f"* separate `--pmc` passes (k_render, per launch): FETCH_SIZE {fetch_kb:.4g} KB, WRITE_SIZE {write_kb:.4g} KB, L2 hit rate {res['L2_hit_rate']:.3f}.\n")
    f.write(f"* HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {traffic/1e9:.1f} GB (uncorrected {(fetch_kb+write_kb)*1024/1e9:.1f} GB) -> {traffic/float(kr['AverageNs']):.1f} GB/s.\n")
    f.write(f"* instruction mix per launch (separate passes): " + ", ".join(f"{k} {res[k]:.4g}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT") if k in res) + "\n")
    f.write(f"* unit figures: " + ", ".join(f"{k} {res[k]:.4g}" for k in ("VALUBusy", "VALUUtilization", "MemUnitBusy", "LDSBankConflict", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if k in res) + "\n")
    f.write(f"* build id (device sources + flags): {res['head']}\n")
    f.write(f"* dispatch: {meta}\n\nbench line under the profiler:\n\n```\n{bench_line}\n```\n")
print(json.dumps(res, indent=1))

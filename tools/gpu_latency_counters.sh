#!/bin/bash
# Where a fetch of the megakernel is served from and how long it takes (separate --pmc passes on the 32-spp probe):
# average VMEM latency (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM), vector-L1 hit rate, L2 hit rate, average latency of the L2's misses to
# the fabric (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ), texture-addresser busy / stalled.   usage (GPU box): tools/gpu_latency_counters.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lat; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
i=0
for grp in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  VMK_NO_TRAV_COUNT=1 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 32 2 > "$OUT/g$i.log" 2>&1 || { echo "group $i ($grp) failed"; tail -2 "$OUT/g$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/g*/*/*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "k_render" in r["Kernel_Name"]]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    big = max(per.values(), key=lambda d: max(d.values()))  # the 32-spp launch (not the 1-spp self check)
    out.update(big)
print(json.dumps(out, indent=1))
g = lambda k: out.get(k, float("nan"))
print("avg VMEM latency (cycles, SQ units)", g("SQ_INST_LEVEL_VMEM") / g("SQ_INSTS_VMEM"))
print("avg LDS latency", g("SQ_INST_LEVEL_LDS") / g("SQ_INSTS_LDS"))
print("vector L1 hit rate", 1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"))
print("L2 hit rate", g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")))
print("avg latency of L2 misses to the fabric (TCC cycles)", g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum"))
print("share of L2 misses that go to DRAM", g("TCC_EA0_RDREQ_DRAM_sum") / g("TCC_EA0_RDREQ_sum"))
PY

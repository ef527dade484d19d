#!/bin/bash
# The cache picture of the TRAVERSAL ALONE (k_trace replay of the megakernel's own rays: no shading, no scratch): vector-L1 and L2 hit
# rates and the latency of L2 misses, to tell how much of the megakernel's miss traffic is the BVH's.  usage: tools/gpu_latency_ktrace.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/latk; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 tools/gpu_replay.py > "$OUT/g$i.log" 2>&1 || { echo "group $i failed"; tail -2 "$OUT/g$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(sys.argv[1] + "/g*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            kind = "shadow" if "Lb1EEv" in r["Kernel_Name"].split("k_trace")[1][:12] and r["Kernel_Name"].split("k_traceILb")[1][4:5] == "1" else "closest"
            tot[kind][r["Counter_Name"]] += float(r["Counter_Value"])
for kind, o in tot.items():
    g = lambda k: o.get(k, float("nan"))
    print(kind, "vector L1 hit %.3f  L2 hit %.3f  L2-miss latency %.0f TCC cycles  (L2 read requests %.3g, fabric reads %.3g)" % (
        1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum"), g("TCP_TCC_READ_REQ_sum"), g("TCC_EA0_RDREQ_sum")))
PY

#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the megakernel on the 32-spp probe (one --pmc pass each): the spill traffic at a glance.
# usage (GPU box): tools/gpu_write_size.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ws; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for c in WRITE_SIZE FETCH_SIZE; do
  VMK_NO_TRAV_COUNT=1 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 32 2 > "$OUT/$c.log" 2>&1
  python3 - "$OUT/$c" $c <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_render" in r["Kernel_Name"]]
per = {}
for r in rows:
    per.setdefault(r["Dispatch_Id"], 0.0); per[r["Dispatch_Id"]] += float(r["Counter_Value"])
big = [v for v in per.values() if v > 0.25 * max(per.values())]
print(sys.argv[2], "per 32-spp launch: %.4g KB  (x8 = %.4g KB per 256 spp); scratch" % (sum(big) / len(big), 8 * sum(big) / len(big)), rows[0]["Scratch_Size"])
PY
done

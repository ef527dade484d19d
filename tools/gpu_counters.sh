#!/bin/bash
# Counter passes for the megakernel (run on the GPU box from the repo root): instruction mix and unit busy figures of an 8-spp
# classroom render, each counter group in its own rocprofv3 run.  Output: gpurun_out/counters/*; summarise with
# tools/pmc_summary.py k_render gpurun_out/counters
ROOT=$(pwd); OUT=$ROOT/gpurun_out/counters
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
B="python3 bench.py --steps 1 --warmup 0 --spp-per-step 8 --no-cpu-baseline --no-replay --no-self-check"
i=0
for grp in "VALUBusy VALUUtilization" "MemUnitBusy LDSBankConflict" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- $B > "$OUT/g$i.log" 2>&1 || { echo "group $i ($grp) failed"; tail -3 "$OUT/g$i.log"; }
done
echo done

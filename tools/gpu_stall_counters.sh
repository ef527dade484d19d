#!/bin/bash
# Issue / wait breakdown of the megakernel (separate --pmc passes, never with a trace): how many wave-cycles wait, for what, and how
# many issue which instruction class.  Results under gpurun_out/stall/; tools/summarize_stall.py prints the per-launch table.
# usage (GPU box, repo root): tools/gpu_stall_counters.sh [spp]
SPP=${1:-64}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stall
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
B="python3 bench.py --config c3 --steps 1 --warmup 1 --spp-per-step $SPP --no-cpu-baseline --no-replay --no-self-check --no-sibling --no-other-configs"
i=0
for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $B > "$OUT/bench_pmc$i.log" 2>&1 || { echo "pmc group $i ($grp) failed"; tail -2 "$OUT/bench_pmc$i.log"; }
  echo "pmc group $i done ($grp)"
done

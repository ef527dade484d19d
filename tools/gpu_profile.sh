#!/bin/bash
# rocprofv3 passes of the bench command (run on the GPU box from the repo root): kernel trace + stats, then separate --pmc
# passes (MI355X_MICROARCH.md: counters in their own runs, never together with a trace): HBM traffic, L2 hits, VALU / SALU /
# LDS / VMEM instruction counts, unit busy figures.  Results under gpurun_out/prof/; tools/summarize_profiles.py <tag> <spp>
# turns them into profiles/<tag>_*.   usage: tools/gpu_profile.sh [spp] [steps] [config]
SPP=${1:-256}; STEPS=${2:-2}; CFG=${3:-c3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
B="python3 bench.py --config $CFG --steps $STEPS --warmup 1 --spp-per-step $SPP --no-cpu-baseline --no-replay --no-self-check --no-sibling --no-other-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $B > "$OUT/bench_kt.log" 2>&1 || exit 1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "VALUBusy VALUUtilization" "MemUnitBusy LDSBankConflict" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $B > "$OUT/bench_pmc$i.log" 2>&1 || { echo "pmc group $i ($grp) failed"; tail -3 "$OUT/bench_pmc$i.log"; }
  echo "pmc group $i done ($grp)"
done
echo done

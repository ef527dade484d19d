#!/bin/bash
# rocprofv3 passes of the bench command (run on the GPU box from the repo root): kernel trace + stats, then three
# separate --pmc passes (MI355X_MICROARCH.md §HBM: counters in their own runs).  Results under gpurun_out/prof/;
# tools/summarize_profiles.py <tag> <spp> turns them into profiles/<tag>_*.   usage: tools/gpu_profile.sh [spp] [steps]
SPP=${1:-256}; STEPS=${2:-2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
B="python3 bench.py --steps $STEPS --warmup 1 --spp-per-step $SPP --no-cpu-baseline --no-replay --no-self-check"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $B > "$OUT/bench_kt.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $B > "$OUT/bench_pmc1.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $B > "$OUT/bench_pmc2.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $B > "$OUT/bench_pmc3.log" 2>&1 || exit 1
echo done

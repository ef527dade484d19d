#!/bin/bash
# How the megakernel's rate scales with resident waves (same binary, same registers: only the grid is capped): 1..6 blocks of 4 waves per CU.
# A rate proportional to the wave count means latency-bound (more parallelism would pay); a flat one means an execution unit is saturated.
# usage (GPU box): tools/gpu_occupancy.sh
for n in 1 2 3 4 5 6; do
  echo "== blocks per CU $n"
  VMK_MAX_BLOCKS_PER_CU=$n VMK_NO_TRAV_COUNT=1 python tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 32 2 2>&1 | grep "^rep 1"
done

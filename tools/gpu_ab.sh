#!/bin/bash
# A/B harness for tuning builds: a 32 spp classroom render (3 repeats, best taken by the reader) for every
# vision_amd/lib/exp/libvmk_*.so variant (VMK_LIB override), interleaved twice to average out clock drift; non-tallying instance.
# usage (on the GPU box): tools/gpu_ab.sh > gpurun_out/ab.log
for pass in 1 2; do
for lib in vision_amd/lib/exp/libvmk_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  VMK_NO_TRAV_COUNT=1 VMK_LIB=$lib timeout -k 10 200 python tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 32 3 2>&1 | grep "^rep [12]" || exit 1
done
done

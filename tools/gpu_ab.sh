#!/bin/bash
# A/B harness for tuning builds: a 32 spp classroom render (3 repeats, best taken by the reader) for the default library and
# every vision_amd/lib/exp/libvmk_*.so variant (VMK_LIB override), interleaved twice to average out clock drift; each library
# is probed with and without the traversal tallies (VMK_NO_TRAV_COUNT).
# usage (on the GPU box): tools/gpu_ab.sh > gpurun_out/ab.log
for pass in 1 2; do
for lib in vision_amd/lib/libvmk.so vision_amd/lib/exp/libvmk_*.so; do
  [ -f "$lib" ] || continue
  for nc in 1; do
  echo "== $lib nocount=$nc"
  VMK_NO_TRAV_COUNT=$nc VMK_LIB=$lib timeout -k 10 200 python tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 32 3 2>&1 | grep "^rep [12]" || exit 1
  done
done
done

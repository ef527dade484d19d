#!/bin/bash
# A/B harness for tuning builds: runs the traversal replay + a 16 spp classroom render for the default library and every
# vision_amd/lib/exp/libvmk_*.so variant (VMK_LIB override).  usage (on the GPU box): tools/gpu_ab.sh > gpurun_out/ab.log
for lib in vision_amd/lib/libvmk.so vision_amd/lib/exp/libvmk_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  VMK_LIB=$lib timeout -k 10 200 python tools/gpu_trace_bench.py 2>&1 | grep -v "^prepare" || exit 1
  VMK_LIB=$lib timeout -k 10 200 python tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 16 2 2>&1 | grep "^rep 1" || exit 1
done

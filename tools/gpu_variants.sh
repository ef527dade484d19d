#!/bin/bash
# For every vision_amd/lib/exp/libvmk_*.so: the cbox_matte megakernel-vs-oracle diagnostic, and (only if that is exact) the
# self-check scenes and a 32 spp classroom perf probe.  usage (GPU box): tools/gpu_variants.sh
for lib in vision_amd/lib/exp/libvmk_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  out=$(VMK_LIB=$lib timeout -k 5 90 python tests/diag_matte.py 2>&1) || { echo "$out" | tail -3; echo "diag failed"; continue; }
  echo "$out" | head -3
  if echo "$out" | grep -q "mismatching pixels 0 of"; then
    VMK_LIB=$lib timeout -k 5 200 python tools/gpu_selfcheck.py 2>&1 | tail -8 || { echo "selfcheck killed"; exit 1; }
    VMK_LIB=$lib timeout -k 10 200 python tools/gpu_perf.py scenes/classroom/vision_scene.json 1920 1080 32 3 2>&1 | grep "^rep [12]" || exit 1
  fi
done

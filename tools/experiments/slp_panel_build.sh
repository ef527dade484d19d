#!/bin/bash
# Builds the SLP-on library variants of profiles/r03_slp_root_cause.md section 1 into vision_amd/lib/exp/ (one codegen knob each;
# tools/gpu_slp_panel.py runs vmk_self_check on cbox_matte for every one of them).  ~4 minutes on 8 cores.
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; D="$ROOT/vision_amd/csrc/device"; T="${TMPDIR:-/tmp}/slp_panel"; mkdir -p "$T" "$ROOT/vision_amd/lib/exp"; cd "$T" || exit 1
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function"   # no -fno-slp-vectorize: that is the point
( /opt/rocm/bin/hipcc $F -fno-slp-vectorize -c "$D/vmk_hero.hip" -o hero.o; /opt/rocm/bin/hipcc $F -fno-slp-vectorize -c "$D/vmk_hero4.hip" -o hero4.o ) &
v() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" -c "$D/vmk.hip" -o "vmk_$name.o" && echo "compiled $name"; }
v plain & v liverange -mllvm -amdgpu-opt-vgpr-liverange=false & v join -mllvm -join-liveintervals=false & wait
v sink -mllvm -disable-machine-sink & v ssc -mllvm -disable-ssc & v fold -mllvm -disable-branch-fold -mllvm -disable-tail-duplicate & v licmcse -mllvm -disable-machine-licm -mllvm -disable-machine-cse & wait
v phisplit -mllvm -phi-elim-split-all-critical-edges & v partial -mllvm -amdgpu-enable-rewrite-partial-reg-uses=false & v peep -mllvm -disable-peephole -mllvm -disable-copyprop & v placement -mllvm -disable-block-placement & wait
v nofuse -mllvm -disable-spill-fusing & v nocopyelim -mllvm -enable-spill-copy-elim=false & v splitedges -mllvm -join-splitedges=false & v regbasic -mllvm -vgpr-regalloc=basic & wait
for o in vmk_*.o; do n=${o#vmk_}; n=${n%.o}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/vision_amd/lib/exp/libvmk_slp_$n.so" "$o" hero.o hero4.o && echo "linked $n"; done

#!/bin/bash
# Build an experimental libvmk variant into vision_amd/lib/exp/libvmk_<name>.so with extra compiler flags (tuning / diagnosis).
# usage: tools/build_variant.sh <name> [extra hipcc flags...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
D="$ROOT/vision_amd/csrc/device"; T="${TMPDIR:-/tmp}/vmk_variants"; mkdir -p "$T" "$ROOT/vision_amd/lib/exp"
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c "$D/vmk.hip" -o "$T/vmk_$name.o" &
/opt/rocm/bin/hipcc $F "$@" -c "$D/vmk_hero.hip" -o "$T/hero_$name.o" &
/opt/rocm/bin/hipcc $F "$@" -c "$D/vmk_hero4.hip" -o "$T/hero4_$name.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/vision_amd/lib/exp/libvmk_$name.so" "$T/vmk_$name.o" "$T/hero_$name.o" "$T/hero4_$name.o"
echo "built vision_amd/lib/exp/libvmk_$name.so"

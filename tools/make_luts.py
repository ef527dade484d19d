#!/usr/bin/env python3
"""Generate the albedo-compensation tables the materials look up at run time (vision_amd/data/luts.bin).

Vision ships these as `base/scattering/precomputed_table.h`, the output of its own `vision-precompute` app
(src/apps/precompute/main.cpp:24-41, Material::precompute_lobe material.h:121-163).  This framework regenerates them
with its own HIP precompute kernel (`vmk_precompute_albedo`, include/vmk.h) instead of copying the reference's numbers;
tests/test_gpu_parity.py checks that kernel bit for bit against the CPU oracle and against a sub-grid of the
reference's tables (tests/golden/lut_subgrid.json) within Monte-Carlo noise, which is what pins the GGX / Fresnel /
sampling code to the reference.  Needs a GPU (run through gpurun); 2^21 samples per texel like the reference take
a few seconds.
Blob layout: u32 magic 'VLUT', u32 version=1, u32 counts[7], then the float tables in vmk_luts order.
"""
import argparse, os, struct, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1 << 21)
    ap.add_argument("--out", default=os.path.join(ROOT, "vision_amd", "data", "luts.bin"))
    a = ap.parse_args()
    from vision_amd import backend
    t0 = time.time()
    tabs = backend.precompute_albedo_tables(a.samples, N)
    print(f"integrated 5 tables in {time.time() - t0:.1f} s; means", [float(t.mean()) for t in tabs])
    sheen = []  # tables 5/6 (sheen LTC coefficients, tools/make_sheen_tables.py) are carried over from the existing blob
    default_blob = os.path.join(ROOT, "vision_amd", "data", "luts.bin")
    if os.path.exists(default_blob):
        raw = open(default_blob, "rb").read()
        _, _, *old = struct.unpack("<II7I", raw[:36])
        off = 36 + 4 * sum(old[:5])
        for c in old[5:]:
            sheen.append(np.frombuffer(raw[off:off + 4 * c], np.float32)); off += 4 * c
    tabs = list(tabs) + sheen
    counts = [len(t) for t in tabs] + [0] * (7 - len(tabs))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "wb") as f:
        f.write(struct.pack("<II7I", 0x54554C56, 1, *counts))
        for t in tabs:
            f.write(np.asarray(t, np.float32).tobytes())
    print("wrote", a.out, os.path.getsize(a.out), "bytes; samples/texel =", a.samples)


if __name__ == "__main__":
    main()

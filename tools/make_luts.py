#!/usr/bin/env python3
"""Generate the albedo-compensation tables the materials look up at run time (vision_amd/data/luts.bin).

Vision ships these as `base/scattering/precomputed_table.h`, the output of its own `vision-precompute` app
(src/apps/precompute/main.cpp:24-41, Material::precompute_lobe material.h:121-163).  This framework regenerates them
with its own lobe code instead of copying the reference's numbers; tests/test_oracle_luts.py then checks the result
against a sub-grid of the reference's tables (tests/golden/lut_subgrid.json) within Monte-Carlo noise, which is what
pins the GGX / Fresnel / sampling code to the reference.

  --backend cpu : integrate with the CPU oracle (bootstrap, moderate sample count)
  --backend gpu : integrate with the HIP precompute kernel of libvmk (vmk_test_eval kind 100), 2^21 samples like the reference
Blob layout: u32 magic 'VLUT', u32 version=1, u32 counts[7], then the float tables in vmk_luts order.
"""
import argparse, os, struct, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 32


def cpu_tables(samples, threads):
    import ctypes as C
    from oracle import oracle_py
    L = oracle_py.lib()
    L.orc_integrate_albedo_table.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    out = []
    for which, count in ((0, N * N), (1, N ** 3 * 2), (2, N ** 3 * 2), (3, N ** 3), (4, N ** 3)):
        t0 = time.time()
        buf = np.zeros(count, np.float32)
        L.orc_integrate_albedo_table(which, N, samples, buf.ctypes.data_as(C.c_void_p), threads)
        print(f"table {which}: {count} floats, {time.time() - t0:.1f} s, mean {buf.mean():.4f}", flush=True)
        out.append(buf)
    return out


def gpu_tables(samples):
    from vision_amd import backend
    return backend.precompute_albedo_tables(samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="cpu", choices=["cpu", "gpu"])
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "vision_amd", "data", "luts.bin"))
    a = ap.parse_args()
    tabs = cpu_tables(a.samples, a.threads) if a.backend == "cpu" else gpu_tables(a.samples)
    counts = [len(t) for t in tabs] + [0, 0]  # sheen LTC tables absent
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "wb") as f:
        f.write(struct.pack("<II7I", 0x54554C56, 1, *counts))
        for t in tabs:
            f.write(np.asarray(t, np.float32).tobytes())
    print("wrote", a.out, os.path.getsize(a.out), "bytes; samples/texel =", a.samples, "backend =", a.backend)


if __name__ == "__main__":
    main()

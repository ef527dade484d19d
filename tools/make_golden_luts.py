#!/usr/bin/env python3
"""Extract a small sub-grid of the reference's own albedo tables as a golden fixture.

Source (read as text, in the build container only): /root/reference/src/base/scattering/precomputed_table.h — the
output of Vision's `vision-precompute` app (2,097,152 samples per texel, produced by the reference's lobe code on a
GPU).  The fixture holds DATA ONLY (indices + values), 8 indices per axis: tests/golden/lut_subgrid.json.
    python tools/make_golden_luts.py
"""
import json, os, re
import numpy as np

SRC = "/root/reference/src/base/scattering/precomputed_table.h"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lut_subgrid.json")
IDX = [0, 4, 9, 13, 18, 22, 27, 31]
N = 32

text = open(SRC).read()
tables = {}
for m in re.finditer(r"const\s+(float2?|float)\s+(\w+)_Table\[(\d+)\]\s*=\s*\{(.*?)\};", text, re.S):
    ty, name, count, body = m.group(1), m.group(2), int(m.group(3)), m.group(4)
    nums = [float(x) for x in re.findall(r"[-+]?\d*\.\d+(?:[eE][-+]?\d+)?|[-+]?\d+(?:[eE][-+]?\d+)?", re.sub(r"float2", "", body))]
    nc = 2 if ty == "float2" else 1
    assert len(nums) == count * nc, (name, len(nums), count, nc)
    tables[name] = np.array(nums, np.float64).reshape(count, nc)

out = {"source": "base/scattering/precomputed_table.h (reference repo), sub-grid indices per axis", "indices": IDX, "tables": {}}
for name, arr in tables.items():
    if arr.shape[0] == N * N:
        sub = [[float(arr[y * N + x, 0]) for x in IDX] for y in IDX]
    else:
        sub = [[[arr[(z * N + y) * N + x].tolist() for x in IDX] for y in IDX] for z in IDX]
    out["tables"][name] = sub
    print(name, arr.shape, "mean", arr.mean())
json.dump(out, open(OUT, "w"))
print("wrote", OUT, os.path.getsize(OUT), "bytes")

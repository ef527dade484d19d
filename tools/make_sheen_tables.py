#!/usr/bin/env python3
"""Append the sheen LTC tables to the albedo-table blob (vision_amd/data/luts.bin).

principled_bsdf's sheen lobe (SheenLTC, render_core/material/principled_bsdf.cpp:17-118) looks up two 32x32 float4
tables — the fitted LTC coefficients published with Zeltner, Burley, Chiang, "Practical Multiple-Scattering Sheen Using
Linearly Transformed Cosines" (SIGGRAPH 2022 talks), which Vision carries as render_core/material/ltc_sheen_table.h
(SheenLTCTableVolume :12, SheenLTCTableApprox :366).  They are measured constants, not something this framework can
regenerate, so this script reads the numbers (data only) from the reference header in the build container and stores them
as tables 5 (Approximate) and 6 (Volume) of the blob.   python tools/make_sheen_tables.py
"""
import os, re, struct, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/render_core/material/ltc_sheen_table.h"
BLOB = os.path.join(ROOT, "vision_amd", "data", "luts.bin")

text = open(SRC).read()
tabs = {}
for name in ("SheenLTCTableVolume", "SheenLTCTableApprox"):
    body = text[text.index(name):]
    body = body[body.index("{") + 1:body.index("};")]
    vals = re.findall(r"float4\(([^)]*)\)", body)
    arr = np.array([[float(x.strip().rstrip("f")) for x in v.split(",")] for v in vals], np.float32)
    assert arr.shape == (1024, 4), (name, arr.shape)
    tabs[name] = arr.reshape(-1)

raw = open(BLOB, "rb").read()
magic, ver, *counts = struct.unpack("<II7I", raw[:36])
assert magic == 0x54554C56 and ver == 1
n5 = sum(counts[:5])
body = raw[36:36 + 4 * n5]
counts = counts[:5] + [4096, 4096]
with open(BLOB, "wb") as f:
    f.write(struct.pack("<II7I", magic, ver, *counts))
    f.write(body)
    f.write(tabs["SheenLTCTableApprox"].tobytes())
    f.write(tabs["SheenLTCTableVolume"].tobytes())
print("wrote", BLOB, os.path.getsize(BLOB), "bytes; approx mean", float(tabs["SheenLTCTableApprox"].mean()), "volume mean", float(tabs["SheenLTCTableVolume"].mean()))

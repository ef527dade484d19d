#!/usr/bin/env python3
"""EXR fixtures from image files the reference ships (data only):
  res/sky.exr (512x256, RGBA half, ZIP)                         -> tests/golden/exr_sky_zip_half.exr   (the file as it is, 110 KB)
  res/render_scene/cbox/TungstenRender.exr (1024x1024, RGB half, PIZ), scanlines 480..543 (two 32-line blocks: the boxes and the walls)
                                                                -> tests/golden/exr_cbox_piz_half.exr  (header rewritten for a 1024x64 window,
                                                                   the two compressed blocks copied verbatim)
  res/render_scene/cbox/TungstenRender.png, the same 64 rows    -> tests/golden/exr_cbox_rows.npy      (uint8: an independent witness of what
                                                                   the PIZ blocks hold — the PNG is that render through a display curve)
  expected statistics of both decodes                           -> tests/golden/exr_expected.json      (written by THIS repo's decoder: a
                                                                   regression pin; the independent checks are the PNG witness and the
                                                                   round trips in tests/test_host.py)
python tools/make_golden_exr.py"""
import json, os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/res"
G = os.path.join(ROOT, "tests", "golden")

open(os.path.join(G, "exr_sky_zip_half.exr"), "wb").write(open(f"{REF}/sky.exr", "rb").read())

b = open(f"{REF}/render_scene/cbox/TungstenRender.exr", "rb").read()
# walk the header, rewriting dataWindow / displayWindow
p = 8; out = bytearray(b[:8]); dw = None
while b[p] != 0:
    e = b.index(b"\0", p); name = b[p:e]; q = e + 1
    e2 = b.index(b"\0", q); typ = b[q:e2]; q = e2 + 1
    size = struct.unpack("<I", b[q:q + 4])[0]; val = b[q + 4:q + 4 + size]
    if name in (b"dataWindow", b"displayWindow"):
        x0, y0, x1, y1 = struct.unpack("<4i", val)
        if name == b"dataWindow": dw = (x0, y0, x1, y1)
        val = struct.pack("<4i", x0, 0, x1, 63)
    out += name + b"\0" + typ + b"\0" + struct.pack("<I", len(val)) + val
    p = q + 4 + size
out += b"\0"; p += 1
h = dw[3] - dw[1] + 1
n_blocks = (h + 31) // 32
offs = struct.unpack(f"<{n_blocks}Q", b[p:p + 8 * n_blocks])
first = 480 // 32
blocks = []
for k in (first, first + 1):
    y, size = struct.unpack("<iI", b[offs[k]:offs[k] + 8])
    assert y == k * 32
    blocks.append(struct.pack("<iI", y - 480, size) + b[offs[k] + 8:offs[k] + 8 + size])
table_at = len(out)
o = table_at + 16
out += struct.pack("<2Q", o, o + len(blocks[0]))
out += blocks[0] + blocks[1]
open(os.path.join(G, "exr_cbox_piz_half.exr"), "wb").write(bytes(out))

from PIL import Image
png = np.asarray(Image.open(f"{REF}/render_scene/cbox/TungstenRender.png").convert("RGB"))[480:544]
np.save(os.path.join(G, "exr_cbox_rows.npy"), png)

from vision_amd.host import load_image
exp = {}
for fn in ("exr_sky_zip_half.exr", "exr_cbox_piz_half.exr"):
    a = load_image(os.path.join(G, fn)).astype(np.float64)
    exp[fn] = {"shape": list(a.shape), "mean": a.mean((0, 1)).tolist(), "max": float(a.max()), "min": float(a.min()),
               "probe": [[int(y), int(x), a[y, x].tolist()] for y, x in ((0, 0), (a.shape[0] // 2, a.shape[1] // 3), (a.shape[0] - 1, a.shape[1] - 1))]}
    print(fn, exp[fn]["shape"], exp[fn]["mean"])
json.dump(exp, open(os.path.join(G, "exr_expected.json"), "w"), indent=1)

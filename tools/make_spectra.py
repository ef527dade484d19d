#!/usr/bin/env python3
"""Write vision_amd/data/spectra.bin: the tabulated physical data the `hero` spectrum mode integrates against.

Contents (all little-endian float32):
  * the CIE 1931 2-degree standard observer x̄, ȳ, z̄ and the CIE standard illuminant D65 (normalised to 1 at 560 nm),
    1 nm steps over 360..830 nm — 471 samples each.  These are published CIE standard tables (CIE 015), not
    reference-authored material; the NUMBERS are read from the reference's header (text only, regex over the
    literals) so that every integral here uses exactly the sample values Vision's `SPD::create_cie_*`
    (src/base/color/spd.cpp:95-115) uses.
  * the measured complex index of refraction (eta, k) of Vision's ten named metals (refractiveindex.info data, as
    tabulated by pbrt and by the reference's material/metal_ior.inl.h), 95 samples each, which `metal` materials
    evaluate per sampled wavelength in hero mode (metal.cpp:113-117, shadernode/spd.cpp:36-39).

Layout: magic 'VSPD', u32 version = 1, u32 n_cie, X[n_cie], Y[n_cie], Z[n_cie], D65[n_cie], u32 n_metals, then per metal
char name[16], u32 n, eta[n], k[n].

Run in the build container (the reference is not present on the GPU box):  python tools/make_spectra.py
"""
import os
import re
import struct

import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vision_amd", "data", "spectra.bin")
NUM = r"[-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?"


def cie_arrays():
    text = open(os.path.join(REF, "base/color/cie.h")).read()
    out = {}
    for name in ("X", "Y", "Z", "D65"):
        m = re.search(r"array<float,\s*cie_sample_count>\s+%s\s*=\s*\{(.*?)\};" % name, text, re.S)
        vals = [float(v) for v in re.findall(NUM, m.group(1).replace("f", " "))]
        assert len(vals) == 471, (name, len(vals))
        out[name] = np.array(vals, np.float32)
    return out


def metals():
    text = open(os.path.join(REF, "render_core/material/metal_ior.inl.h")).read()
    out = []
    for m in re.finditer(r"ocarina::array\s+(\w+)\s*\{(.*?)\};", text, re.S):
        pairs = re.findall(r"make_float2\(\s*(%s)f?\s*,\s*(%s)f?\s*\)" % (NUM, NUM), m.group(2))
        out.append((m.group(1), np.array([float(a) for a, _ in pairs], np.float32), np.array([float(b) for _, b in pairs], np.float32)))
    return out


def main():
    cie = cie_arrays()
    mt = metals()
    with open(OUT, "wb") as f:
        f.write(b"VSPD" + struct.pack("<II", 1, 471))
        for k in ("X", "Y", "Z", "D65"):
            f.write(cie[k].tobytes())
        f.write(struct.pack("<I", len(mt)))
        for name, eta, k in mt:
            f.write(name.encode().ljust(16, b"\0") + struct.pack("<I", len(eta)) + eta.tobytes() + k.tobytes())
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(mt), "metals:", " ".join(n for n, _, _ in mt))
    print("Y[195] (555 nm) =", cie["Y"][195], " D65[200] (560 nm) =", cie["D65"][200])


if __name__ == "__main__":
    main()

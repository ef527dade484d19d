"""Pin the CPU oracle (no GPU): known-answer vectors derived from the reference's closed forms and a sub-grid of the
reference's own precomputed albedo tables (tests/golden/lut_subgrid.json, extracted by tools/make_golden_luts.py
from base/scattering/precomputed_table.h — the output of the reference's lobe code, 2^21 samples per texel)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, f2u
from oracle import oracle_py


def tea_np(v0, v1, rounds=4):
    """math/util.h:13-24 restated with numpy uint32 arithmetic."""
    v0 = np.uint32(v0); v1 = np.uint32(v1); s0 = np.uint32(0)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            s0 = np.uint32(s0 + np.uint32(0x9e3779b9))
            v0 = np.uint32(v0 + (np.uint32(np.uint32(v1 << np.uint32(4)) + np.uint32(0xa341316c)) ^ np.uint32(v1 + s0) ^ np.uint32(np.uint32(v1 >> np.uint32(5)) + np.uint32(0xc8013ea4))))
            v1 = np.uint32(v1 + (np.uint32(np.uint32(v0 << np.uint32(4)) + np.uint32(0xad90777d)) ^ np.uint32(v0 + s0) ^ np.uint32(np.uint32(v0 >> np.uint32(5)) + np.uint32(0x7e95761e))))
    return v0


def test_tea_lcg_known_answers(built):
    rng = np.random.default_rng(0)
    v = rng.integers(0, 2 ** 32, (64, 2), dtype=np.uint64).astype(np.uint32)
    out = oracle_py.test_eval_noscene(51, v.view(np.float32), 1).view(np.uint32)[:, 0]
    assert [int(x) for x in out] == [int(tea_np(a, b)) for a, b in v]
    # sampler: state = tea(tea(px,py), tea(frame,dim)); u = (lcg(state) & 0xffffff) / 2^24  (independent.cpp:24-27, util.h:27-33)
    q = np.array([[3, 7, 0, 0], [1919, 1079, 1023, 1], [0, 0, 5, 0xFFFFFFFF]], np.uint32)
    got = oracle_py.test_eval_noscene(0, q.view(np.float32), 8)
    for row, g in zip(q, got):
        st = int(tea_np(tea_np(row[0], row[1]), tea_np(row[2], row[3])))
        exp = []
        for _ in range(8):
            st = (1664525 * st + 1013904223) & 0xFFFFFFFF
            exp.append(np.float32(st & 0xFFFFFF) * np.float32(1.0 / 16777216.0))
        assert np.array_equal(g, np.array(exp, np.float32))


def test_elementary_functions_accuracy(built):
    """The deterministic sin/cos/acos/atan2/exp/log kernels stay within 2.5 ulp (atan2: 4 ulp) of double-precision libm on their domains."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-7, 7, 20000), np.linspace(-1, 1, 2001)]).astype(np.float32)
    y = rng.uniform(-3, 3, x.shape[0]).astype(np.float32)
    out = oracle_py.test_eval_noscene(1, np.stack([x, y], 1), 7)
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    lx = (np.abs(x) * np.float32(0.125) + np.float32(5.9604645e-8)).astype(np.float64)  # the unit's log argument, (0, 1)
    ref = [np.sin(xd), np.cos(xd), np.arccos(np.clip(xd, -1, 1)), np.arctan2(yd, xd), np.exp(-np.abs(xd)), np.sqrt(np.abs(xd)), np.log(lx)]
    for k, r in enumerate(ref):
        ulp = np.spacing(np.maximum(np.abs(r), 1e-3).astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(out[:, k].astype(np.float64) - r) / ulp) <= (4.0 if k == 3 else 2.5), k


def test_refract_vector_of_reference_test_bxdf(built):
    """src/tests/test_bxdf.cpp:28-40 refracts wo = normalize(1,0,-2) about +-z with eta = 1.5 (prints, no expected value);
    expected values here come from optics.h:28-39 evaluated in float64."""
    wo = np.array([1.0, 0.0, -2.0]); wo /= np.linalg.norm(wo)
    rows, exp = [], []
    for n in ([0, 0, 1.0], [0, 0, -1.0]):
        n = np.array(n)
        rows.append(np.concatenate([wo, n, [1.5]]))
        ci = n @ wo
        st2 = max(0.0, 1 - ci * ci) / 1.5 ** 2
        ct = np.sqrt(max(0.0, 1 - st2))
        exp.append(np.concatenate([[1.0 if st2 < 1 else 0.0], -wo / 1.5 + (ci / 1.5 - ct) * n]))
    out = oracle_py.test_eval_noscene(50, np.array(rows, np.float32), 7)
    assert np.allclose(out[:, :4], np.array(exp), atol=2e-7)
    c = abs(wo[2])
    st2 = (1 - c * c) / 1.5 ** 2; ct = np.sqrt(1 - st2)
    F = 0.5 * (((1.5 * c - ct) / (1.5 * c + ct)) ** 2 + ((c - 1.5 * ct) / (c + 1.5 * ct)) ** 2)
    assert abs(out[0, 4] - F) < 1e-6 and abs(out[0, 6] - 0.04) < 1e-7


LUT_NAMES = ["PureReflectionLobe", "DielectricLobe", "DielectricInvLobe", "SpecularLobe", "CoatLobe"]


@pytest.mark.parametrize("which", range(5))
def test_albedo_tables_reproduce_the_reference(built, which):
    """Re-integrate texels of the reference's precomputed tables with the oracle's own lobe code (2^16 samples; the
    reference used 2^21).  Agreement within Monte-Carlo noise pins GGX D/G/VNDF sampling, both PDFs, BRDF/BTDF,
    dielectric/Schlick Fresnel and the reflect/transmit selection end to end."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lut_subgrid.json")))
    idx = gold["indices"]
    ref = np.array(gold["tables"][LUT_NAMES[which]], np.float64)
    samples = 1 << 16
    rng = np.random.default_rng(which)
    picks = [(rng.integers(0, 8), rng.integers(0, 8), rng.integers(0, 8)) for _ in range(10)]
    worst = 0.0
    for (ix, iy, iz) in picks:
        x, y, z = idx[ix], idx[iy], idx[iz]
        if which == 0:
            got = oracle_py.integrate_albedo(0, 32, x, y, 0, samples)[:1]
            want = ref[iy, ix].reshape(1)
        else:
            got = oracle_py.integrate_albedo(which, 32, x, y, z, samples)
            want = ref[iz, iy, ix].reshape(-1)
            got = got[: want.shape[0]]
        err = np.abs(got.astype(np.float64) - want)
        tol = 0.02 * np.maximum(np.abs(want), 0.05) + 0.004  # ~5 sigma of a 2^16-sample estimate of O(1) throughput weights
        worst = max(worst, float((err / tol).max()))
        assert (err <= tol).all(), (LUT_NAMES[which], (x, y, z), got, want)
    assert worst > 0.0


def test_shipped_lut_blob_matches_reference_tables(built):
    """The run-time tables (vision_amd/data/luts.bin, generated by this repo's own precompute) agree with the
    reference's tables on the golden sub-grid."""
    import struct
    raw = open(os.path.join(ROOT, "vision_amd", "data", "luts.bin"), "rb").read()
    magic, ver, *counts = struct.unpack("<II7I", raw[:36])
    assert magic == 0x54554C56 and ver == 1
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lut_subgrid.json")))
    idx = gold["indices"]; N = 32; off = 36
    for i, name in enumerate(LUT_NAMES):
        t = np.frombuffer(raw[off:off + 4 * counts[i]], np.float32); off += 4 * counts[i]
        ref = np.array(gold["tables"][name], np.float64)
        if i == 0:
            mine = np.array([[t[y * N + x] for x in idx] for y in idx])
        else:
            nc = 2 if i in (1, 2) else 1
            tt = t.reshape(N, N, N, nc)
            mine = np.array([[[tt[z, y, x] for x in idx] for y in idx] for z in idx]).reshape(ref.shape)
        assert np.mean(np.abs(mine - ref)) < 0.01 and np.max(np.abs(mine - ref)) < 0.06, name

"""GPU parity tests (run on a real MI355X with -m gpu).  Everything goes through the C-ABI (libvmk.so); the CPU
oracle is only the checker.  Bar: bit-exact for the integer paths (RNG, hit ids, counters); float32 radiance within
the north-star tolerance of 1e-4 relative L2 (the build is designed to be bit-exact, and the tests report it)."""
import os

import numpy as np
import pytest

from conftest import ROOT, f2u, rel_l2

pytestmark = pytest.mark.gpu

TOL_REL_L2 = 1e-4  # BASELINE.json north_star: per-pixel radiance <= 1e-4 relative L2


@pytest.fixture(scope="module")
def backend(built):
    import ctypes
    from vision_amd.backend import Backend, lib_path
    assert os.path.exists(lib_path()), "HIP extension missing on the GPU box — no fallback exists"
    be = Backend(0)
    yield be
    be.close()


def _load(be, rel_path, w, h, **kw):
    from vision_amd.host import HostScene
    from oracle import oracle_py
    hs = HostScene(os.path.join(ROOT, rel_path), width=w, height=h, **kw)
    p = hs.params_copy()
    be.upload_scene(hs)
    info = be.build_accel()
    be.set_render_params(p)
    return hs, p, oracle_py.OracleScene(hs), info


def _bits_equal(a, b):
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def test_device_units_rng_math_warps_microfacet(backend):
    from oracle import oracle_py
    rng = np.random.default_rng(7)
    q = np.stack([rng.integers(0, 4096, 4096), rng.integers(0, 4096, 4096), rng.integers(0, 1 << 20, 4096), rng.integers(0, 2, 4096) * 0xFFFFFFFF], 1).astype(np.uint32)
    assert np.array_equal(backend.test_eval(0, q.view(np.float32), 8), oracle_py.test_eval_noscene(0, q.view(np.float32), 8))
    x = np.stack([np.concatenate([rng.uniform(-7, 7, 8000), np.linspace(-1, 1, 192)]), rng.uniform(-3, 3, 8192)], 1).astype(np.float32)
    assert _bits_equal(backend.test_eval(1, x, 7), oracle_py.test_eval_noscene(1, x, 7))
    u = rng.uniform(0, 1, (8192, 2)).astype(np.float32)
    u[:4] = [[0, 0], [0, 0.5], [0.999999, 0.999999], [0.5, 0]]
    assert _bits_equal(backend.test_eval(2, u, 8), oracle_py.test_eval_noscene(2, u, 8))
    a = np.concatenate([rng.normal(size=(8192, 3)), rng.uniform(0, 1, (8192, 2)), rng.uniform(0.001, 1, (8192, 2)), rng.uniform(1.01, 3, (8192, 1))], 1).astype(np.float32)
    assert _bits_equal(backend.test_eval(3, a, 8), oracle_py.test_eval_noscene(3, a, 8))


@pytest.mark.parametrize("scene", ["scenes/cbox/cbox_matte.json", "scenes/cbox/cbox_materials.json", "scenes/cbox/cbox_media.json", "scenes/cbox/cbox_sheen.json", "scenes/cbox/cbox_extra.json",
                                   "scenes/cbox/cbox_normal.json"])
def test_material_camera_and_path_units(backend, scene):
    """Every material type: evaluate + sample on random (wo, wi, uv, rng stream); camera rays; whole-path records."""
    hs, p, osc, _ = _load(backend, scene, 48, 48, mediums=scene.endswith("cbox_media.json"))
    rng = np.random.default_rng(11)
    nm = hs.scene.n_materials
    n = 400 * nm
    a = np.zeros((n, 12), np.float32)
    a[:, 0] = f2u(np.repeat(np.arange(nm), 400)); a[:, 1] = f2u(rng.integers(0, 64, n)); a[:, 2] = f2u(rng.integers(0, 64, n)); a[:, 3] = f2u(rng.integers(0, 64, n))
    a[:, 4:7] = rng.normal(size=(n, 3)); a[:, 7:10] = rng.normal(size=(n, 3)); a[:, 10:12] = rng.uniform(-1, 2, (n, 2))
    a[::7, 6] = 0.0  # grazing wo
    assert _bits_equal(backend.test_eval(4, a, 13), osc.test_eval(p, 4, a, 13))
    pix = np.stack([rng.integers(0, 48, 2048), rng.integers(0, 48, 2048), rng.integers(0, 1024, 2048)], 1).astype(np.uint32).view(np.float32)
    assert _bits_equal(backend.test_eval(5, pix, 6), osc.test_eval(p, 5, pix, 6))
    yy, xx = np.mgrid[0:48, 0:48]
    pix6 = np.stack([xx.ravel(), yy.ravel(), np.full(48 * 48, 3)], 1).astype(np.uint32).view(np.float32)
    assert _bits_equal(backend.test_eval(6, pix6, 67), osc.test_eval(p, 6, pix6, 67))


@pytest.mark.parametrize("scene, w, h", [("scenes/cbox/cbox_materials.json", 64, 64), ("scenes/classroom/vision_scene.json", 64, 36)])
def test_traversal_hits_match_oracle(backend, scene, w, h):
    """GPU LBVH + LDS-stack traversal vs the oracle's own BVH: identical (inst, prim, bary) and occlusion bits."""
    hs, p, osc, info = _load(backend, scene, w, h)
    assert info["node_bytes"] == 128 and info["tri_bytes"] == 48
    rng = np.random.default_rng(5)
    lo, hi = np.array(list(hs.scene.world_min)), np.array(list(hs.scene.world_max))
    n = 20000
    org = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs[:100, 0] = 0.0; dirs[100:200, 1] = 0.0; dirs[200:300, :2] = 0.0  # axis-parallel rays (inf / NaN slabs)
    tmax = np.full(n, 3.0e38, np.float32)
    hg, _ = backend.trace(org, dirs, tmax)
    ho = osc.trace(org, dirs, tmax)
    assert np.array_equal(hg, ho)
    assert (hg[:, 0] != 0xFFFFFFFF).mean() > 0.3
    tm = rng.uniform(0.05, 3.0, n).astype(np.float32)
    og, _ = backend.trace(org, dirs, tm, any_hit=True)
    oo = osc.trace(org, dirs, tm, any_hit=True)
    assert np.array_equal(og[:, 0], oo[:, 0])


@pytest.mark.parametrize("name, scene, w, h, spp", [
    ("cbox_matte", "scenes/cbox/cbox_matte.json", 32, 32, 8),
    ("cbox_materials", "scenes/cbox/cbox_materials.json", 32, 32, 4),
    ("classroom", "scenes/classroom/vision_scene.json", 48, 27, 2),
    ("cbox_media", "scenes/cbox/cbox_media.json", 32, 32, 4),          # homogeneous media + HG phase function (§8f-1)
    ("classroom_fog", "scenes/classroom/vision_scene.json", 48, 27, 2),  # the scene as shipped: global fog
    ("cbox_lights", "scenes/cbox/cbox_lights.json", 32, 32, 4),         # point + spot lights, mitchell filter (§8f-2)
    ("cbox_sinc", "scenes/cbox/cbox_sinc.json", 32, 32, 4),             # point light, Lanczos-sinc filter
    ("glass_of_water", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),  # config 4 (srgb): glass + metal, depth 32
    ("cbox_power", "scenes/cbox/cbox_power.json", 32, 32, 4),           # lightsampler/power
    ("cbox_sheen", "scenes/cbox/cbox_sheen.json", 32, 32, 4),           # principled_bsdf sheen (LTC) layer
    ("cbox_extra", "scenes/cbox/cbox_extra.json", 32, 32, 4),           # material/metallic, material/add
    ("cbox_normal", "scenes/cbox/cbox_normal.json", 32, 32, 4),         # normal-mapped shading frames, mix/add of principled + single lobe, shape/sphere
    ("playground", "scenes/playground/vision_scene.json", 48, 48, 2),   # the reference's playground scene: "multiply" node (image x constant), normal map, mix, principled
    ("staircase2", "scenes/staircase2/vision_scene.json", 48, 27, 2),   # a complete shipped scene: 19 meshes, 13 area lights, JPG textures
    ("cbox_vision_scene", "scenes/cbox/vision_scene.json", 48, 48, 2),  # the reference's default Cornell scene: spot, point, spherical, projector and area lights
    # spectrum/hero (§8f-2): the vmk_hero.hip instance of the megakernel, all four <FULL, MEDIA> variants
    ("cbox_hero", "scenes/cbox/cbox_hero.json", 32, 32, 4),             # every material family, dispersive BK7 glass, measured Cu, texture
    ("cbox_hero_matte", "scenes/cbox/cbox_hero_matte.json", 32, 32, 4), # single-lobe variant
    ("cbox_hero_media", "scenes/cbox/cbox_hero_media.json", 32, 32, 4), # sigma_a / sigma_s uplifted as unbound spectra
    ("cbox_hero_lights", "scenes/cbox/cbox_hero_lights.json", 32, 32, 4),  # point + spot illumination spectra
    ("glass_of_water_hero", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),  # config 4 as worded: "spectral glass"
    ("classroom_hero", "scenes/classroom/vision_scene.json", 48, 27, 2),            # environment map + image textures through the uplift
    # spectrum/hero, "dimension": 4 — the vmk_hero4.hip instance (four wavelengths per path) against the ORC_SPEC_DIM = 4 oracle build
    ("cbox_hero4", "scenes/cbox/cbox_hero.json", 32, 32, 4),
    ("cbox_hero4_matte", "scenes/cbox/cbox_hero_matte.json", 32, 32, 4),
    ("cbox_hero4_media", "scenes/cbox/cbox_hero_media.json", 32, 32, 4),
    ("glass_of_water_hero4", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),
    ("cbox_prism_hero4", "scenes/cbox/cbox-prism.json", 48, 48, 2),  # the reference's own "dimension": 4 scene as shipped (dispersive sphere, checker.jpg)
])
def test_render_matches_oracle_and_golden(backend, name, scene, w, h, spp):
    hero = "hero" in name
    hs, p, osc, _ = _load(backend, scene, w, h, mediums=name in ("cbox_media", "classroom_fog", "cbox_hero_media", "cbox_hero4_media"),
                          spectrum="hero4" if "hero4" in name else ("hero" if name in ("glass_of_water_hero", "classroom_hero") else None))
    assert (hs.scene.spectrum == 1) == hero and hs.scene.spectrum_dimension == (4 if "hero4" in name else 3)
    # the megakernel variant this scene selects agrees with the unit kernel (vmk_self_check); hero scenes have their own
    # instance of the path unit kernel (vmk_hero.hip k_unit_path)
    assert backend.self_check() == w * h
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, spp)
    img = backend.download_accum()
    cg = backend.counters()
    ref, co = osc.render(p, 0, spp)
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}_{w}x{h}x{spp}.npy"))
    assert np.isfinite(img).all()
    assert rel_l2(img, ref) <= TOL_REL_L2 and rel_l2(img, gold) <= TOL_REL_L2
    # integer side of the path is exact: same number of rays, hits and paths
    for k in ("closest_rays", "shadow_rays", "paths", "surface_hits"):
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "not bit-exact (within tolerance, but the build is designed to be exact)"
    if hero:  # the hero unit kernel also reproduces the oracle's per-vertex path records
        yy, xx = np.mgrid[0:h, 0:w]
        pix6 = np.stack([xx.ravel(), yy.ravel(), np.full(w * h, 1)], 1).astype(np.uint32).view(np.float32)
        assert _bits_equal(backend.test_eval(6, pix6, 67), osc.test_eval(p, 6, pix6, 67))


def test_config1_cbox_1024x1024_matches_oracle_on_sampled_tiles(backend):
    """BASELINE.json configs[1] at its full resolution (cbox 1024x1024, matte-only): the GPU renders the whole image, the
    oracle every 64th 32x32 tile of it (tile ownership shards exactly), and those 16k pixels agree bit for bit."""
    from vision_amd import _abi
    hs, p, osc, _ = _load(backend, "scenes/cbox/cbox_matte.json", 1024, 1024)
    backend.reset_accum()
    backend.render_batch(0, 8)
    img = backend.download_accum()
    ref, cc = osc.render(p, 0, 8, tiles=_abi.Tiles(32, 5, 64))
    owned = ref[..., 3] != 0.0
    assert owned.sum() == 16 * 32 * 32 and cc["paths"] == owned.sum() * 8
    assert np.array_equal(img[owned].view(np.uint32), ref[owned].view(np.uint32))
    assert np.isfinite(img).all() and (img[..., 3] == 1.0).all()


@pytest.mark.parametrize("scene, w, h, frames, kw", [
    ("scenes/classroom/vision_scene.json", 1920, 1080, 2, {}),                 # hero <false, false> at the headline size
    ("scenes/classroom/vision_scene.json", 960, 540, 2, {"mediums": True}),    # hero <false, true>: the scene's global fog
    ("scenes/glass-of-water/vision_scene.json", 1024, 1024, 2, {}),            # config 4 "spectral glass" at its resolution
    # BASELINE config 4 exactly as worded: 1024x1024, spectrum hero, max depth 64 (+ min depth 3) — the divergence stress
    ("scenes/glass-of-water/vision_scene.json", 1024, 1024, 2, {"max_depth": 64, "min_depth": 3}),
    # the four-wavelength instance (vmk_hero4.hip) at full size: classroom at the headline resolution, config 4 at depth 64, and the
    # reference's own dimension-4 scene at the 1024^2 its JSON asks for
    ("scenes/classroom/vision_scene.json", 1920, 1080, 2, {"spectrum": "hero4"}),
    ("scenes/glass-of-water/vision_scene.json", 1024, 1024, 2, {"max_depth": 64, "min_depth": 3, "spectrum": "hero4"}),
    ("scenes/cbox/cbox-prism.json", 1024, 1024, 2, {"spectrum": None}),
])
def test_hero_full_size_matches_oracle_on_sampled_tiles(backend, scene, w, h, frames, kw):
    """The hero-spectrum instances of the megakernel (vmk_hero.hip, vmk_hero4.hip) at full size: vmk_self_check against the unit-kernel
    twin, then the GPU renders the whole image, the oracle every 64th 32x32 tile, bit for bit."""
    from vision_amd import _abi
    kw = dict(kw)
    spectrum = kw.pop("spectrum", "hero")
    hs, p, osc, _ = _load(backend, scene, w, h, spectrum=spectrum, **kw)
    assert hs.scene.spectrum == 1 and hs.scene.spectrum_dimension == (3 if spectrum == "hero" else 4)
    assert backend.self_check() > 4000
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, frames)
    img = backend.download_accum()
    ref, cc = osc.render(p, 0, frames, tiles=_abi.Tiles(32, 7, 64))
    owned = ref[..., 3] != 0.0
    assert owned.sum() > 0 and cc["paths"] == owned.sum() * frames
    assert np.array_equal(img[owned].view(np.uint32), ref[owned].view(np.uint32))
    assert np.isfinite(img).all() and (img[..., 3] == 1.0).all()


def test_staircase_as_shipped_matches_oracle_on_sampled_tiles(backend):
    """The reference's staircase scene exactly as shipped (263 k triangles in 775 OBJ meshes, 9 JPG textures, substrate / metal / glass /
    diffuse, one area light; no stand-in) at its own 720 x 1280: the self check, and every 97th 32 x 32 tile of two frames against the
    oracle, bit for bit.  (Its gallery picture was taken from another camera than the file's: DESIGN.md section 2.)"""
    from vision_amd import _abi
    hs, p, osc, info = _load(backend, "scenes/staircase/vision_scene.json", 720, 1280)
    assert hs.scene.n_tris > 260000 and "stand-in" not in hs.description
    assert backend.self_check() > 4000
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, 2)
    full = backend.download_accum()
    ref, cc = osc.render(p, 0, 2, tiles=_abi.Tiles(32, 5, 97))
    owned = ref[..., 3] != 0.0
    assert owned.sum() >= 8 * 32 * 16 and np.isfinite(full).all()
    assert np.array_equal(full[owned].view(np.uint32), ref[owned].view(np.uint32))


def test_glass_of_water_depth_64_parity(backend):
    """BASELINE config 4's integrator setting (max depth 64, min depth 3) on the glass-of-water scene in srgb mode."""
    hs, p, osc, _ = _load(backend, "scenes/glass-of-water/vision_scene.json", 96, 96, max_depth=64, min_depth=3)
    assert p.max_depth == 64
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, 2)
    img = backend.download_accum()
    ref, co = osc.render(p, 0, 2)
    cg = backend.counters()
    assert rel_l2(img, ref) <= TOL_REL_L2
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert cg["closest_rays"] == co["closest_rays"] and cg["shadow_rays"] == co["shadow_rays"]


def test_larger_render_parity_classroom(backend):
    """Textured + environment-lit scene with metal / glass / substrate at a size the oracle still finishes in seconds."""
    hs, p, osc, _ = _load(backend, "scenes/classroom/vision_scene.json", 160, 90)
    backend.reset_accum()
    backend.render_batch(0, 4)
    img = backend.download_accum()
    ref, _ = osc.render(p, 0, 4)
    assert rel_l2(img, ref) <= TOL_REL_L2
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "within tolerance but not bit-exact: the build is designed to be exact"
    tm = backend.tonemap(final_picture=True)
    from oracle import oracle_py
    assert np.abs(tm - oracle_py.tonemap(p, ref, True)).max() < 2e-6  # sRGB pow differs in the last ulp (display path only)


def test_full_size_properties(backend):
    """Size-independent properties at BASELINE.json's full classroom size (1920x1080), few frames:
    determinism, batch-split invariance, tile-sharding linearity, consistent ray accounting."""
    from vision_amd import _abi
    hs, p, osc, info = _load(backend, "scenes/classroom/vision_scene.json", 1920, 1080)
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, 2)
    a = backend.download_accum()
    c = backend.counters()
    assert np.isfinite(a).all() and (a[..., 3] == 1.0).all()
    assert c["paths"] == 1920 * 1080 * 2 and c["shadow_rays"] == c["surface_hits"] and c["closest_rays"] >= c["paths"]
    backend.reset_accum()
    backend.render_batch(0, 1); backend.render_batch(1, 1)
    b = backend.download_accum()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # 2 frames == 1 + 1 frames, run-to-run deterministic
    total = np.zeros_like(a)
    for rank in range(2):
        backend.reset_accum()
        backend.render_batch(0, 2, tiles=_abi.Tiles(32, rank, 2))
        total += backend.download_accum()
    assert np.array_equal(total.view(np.uint32), a.view(np.uint32))  # disjoint tiles: sum over ranks is exact
    # spot-check 256 random pixels of the full-size frame against the oracle's per-pixel path records
    rng = np.random.default_rng(3)
    pix = np.stack([rng.integers(0, 1920, 256), rng.integers(0, 1080, 256), np.zeros(256)], 1).astype(np.uint32).view(np.float32)
    assert _bits_equal(backend.test_eval(6, pix, 67), osc.test_eval(p, 6, pix, 67))


def test_error_behaviour(backend):
    """C-ABI error convention: negative status + message, no crash (the reference logs OC_ERROR and aborts)."""
    import ctypes as C
    from vision_amd.backend import Backend, BackendError
    be = Backend(0)
    with pytest.raises(BackendError, match="not ready"):
        be.render_batch(0, 1)
    with pytest.raises(BackendError, match="bad argument"):
        be.test_eval(99, np.zeros((1, 4), np.float32), 8)
    # tabulated-spectrum ("spd") slots and the uplift table belong to the hero spectrum: inconsistent tables are refused on the
    # host, before anything reaches a kernel
    from vision_amd.host import HostScene
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_hero.json"), width=16, height=16)
    hs.scene.spectrum = 0   # claim sRGB while the metal / glass slots still reference spectra
    with pytest.raises(BackendError, match="missing texture or spectrum"):
        be.upload_scene(hs)
    hs.scene.spectrum = 1
    keep = hs.scene.spd_cie_count
    hs.scene.spd_cie_count = 1 << 20   # CIE tables reaching past the spectra pool
    with pytest.raises(BackendError, match="CIE tables out of range"):
        be.upload_scene(hs)
    hs.scene.spd_cie_count = keep
    be.upload_scene(hs)      # the untouched tables are accepted
    be.build_accel(); be.set_render_params(hs.params_copy())
    assert np.isfinite(be.render_aov(0)["albedo"]).all()  # the G-buffer pass exists for every spectrum instance
    with pytest.raises(BackendError, match="sRGB instance"):  # the ray capture and the material unit kernel do not
        be.test_eval(7, np.zeros((1, 3), np.float32), 1 + 16 * 24)
    be.close()


@pytest.mark.parametrize("scene,w,h", [("scenes/cbox/cbox_materials.json", 40, 40), ("scenes/classroom/vision_scene.json", 64, 36)])
def test_captured_rays_match_oracle_and_replay(backend, scene, w, h):
    """The ray capture that feeds the traversal replay (k_test kind 7) returns exactly the rays the oracle's Li()
    traces — origins, directions and t_max bit for bit — and replaying them through k_trace gives the oracle's hits."""
    hs, p, osc, _ = _load(backend, scene, w, h)
    yy, xx = np.mgrid[0:h, 0:w]
    pix = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.uint32)
    g = backend.capture_rays(pix, frame=3)
    o = osc.dump_rays(p, frame=3, stride=1)
    # oracle order is (path, seq); vertex = closest rays traced so far on the path - 1; GPU order is (vertex, kind, path)
    first = np.r_[True, o["path"][1:] != o["path"][:-1]]
    closest_cum = np.cumsum(o["kind"] == 0)
    base = np.maximum.accumulate(np.where(first, closest_cum - (o["kind"] == 0), 0))
    vertex = closest_cum - base - 1
    assert vertex.max() < 24, "capture holds 24 vertices per path"
    order = np.lexsort((o["path"], o["kind"], vertex))
    assert g["kind"].shape == o["kind"].shape
    assert np.array_equal(g["kind"], o["kind"][order])
    for k in ("org", "dir", "tmax"):
        assert _bits_equal(g[k], o[k][order]), k
    for kind in (0, 1):
        m = g["kind"] == kind
        hg, _ = backend.trace(g["org"][m], g["dir"][m], g["tmax"][m], any_hit=bool(kind))
        ho = osc.trace(g["org"][m], g["dir"][m], g["tmax"][m], any_hit=bool(kind))
        assert np.array_equal(hg, ho)


@pytest.mark.parametrize("which", range(5))
def test_hip_albedo_precompute_matches_oracle_and_reference_tables(backend, which):
    """vmk_precompute_albedo (the HIP version of the reference's vision-precompute app) is bit-identical to the oracle's
    integrator on a small grid, and at 2^18 samples per texel it reproduces the reference's shipped tables
    (tests/golden/lut_subgrid.json) within Monte-Carlo noise."""
    import ctypes as C
    import json
    from oracle import oracle_py
    res, samples = 6, 192
    got = backend.precompute_albedo(which, res, samples)
    want = np.zeros_like(got)
    L = oracle_py.lib()
    L.orc_integrate_albedo_table.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    L.orc_integrate_albedo_table(which, res, samples, want.ctypes.data_as(C.c_void_p), 4)
    assert _bits_equal(got, want)
    names = ["PureReflectionLobe", "DielectricLobe", "DielectricInvLobe", "SpecularLobe", "CoatLobe"]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lut_subgrid.json")))
    idx, N = gold["indices"], 32
    ref = np.array(gold["tables"][names[which]], np.float64)
    t = backend.precompute_albedo(which, N, 1 << 18)
    if which == 0:
        mine = np.array([[t[y * N + x] for x in idx] for y in idx])
    else:
        tt = t.reshape(N, N, N, 2 if which in (1, 2) else 1)
        mine = np.array([[[tt[z, y, x] for x in idx] for y in idx] for z in idx]).reshape(ref.shape)
    err = np.abs(mine - ref)
    tol = 0.02 * np.maximum(np.abs(ref), 0.05) + 0.004
    if which != 0:
        tol[0] *= 2.0  # z = 0 is the index-matched end (ior 1.003): heavy-tailed weights, the reference's own 2^21-sample texels scatter by +-0.02 there
    assert (err <= tol).all(), (names[which], float((err / tol).max()))


@pytest.mark.parametrize("scene, w, h, kw", [("scenes/cbox/cbox_materials.json", 64, 64, {}), ("scenes/classroom/vision_scene.json", 96, 54, {}), ("scenes/cbox/cbox_sheen.json", 48, 48, {}), ("scenes/cbox/cbox_extra.json", 48, 48, {}),
                                            ("scenes/cbox/cbox_lights.json", 40, 40, {}), ("scenes/cbox/cbox_normal.json", 48, 48, {}),
                                            # spectrum/hero: albedo and emission are spectra brought to linear sRGB through the pixel's wavelengths
                                            ("scenes/cbox/cbox_hero.json", 48, 48, {}), ("scenes/cbox/cbox_hero.json", 48, 48, {"spectrum": "hero4"}),
                                            ("scenes/cbox/cbox-prism.json", 40, 40, {}), ("scenes/classroom/vision_scene.json", 96, 54, {"spectrum": "hero"})])
def test_aov_planes_match_oracle(backend, scene, w, h, kw):
    """vmk_render_aov (the reference's G-buffer kernel, frame_buffer.cpp:156-219): shading normal, linear depth, material
    albedo (every lobe class incl. the LUT-based coat / specular albedos of principled_bsdf) and emission, bit for bit — for all
    three spectrum instances (sRGB; hero with three and four wavelengths: `linear_srgb(bsdf.albedo(wo), swl)`, :192-203)."""
    hs, p, osc, _ = _load(backend, scene, w, h, **kw)
    g = backend.render_aov(frame=2)
    o = osc.render_aov(p, frame=2)
    for k in ("normal", "albedo", "emission"):
        assert _bits_equal(g[k], o[k]), k
    assert np.allclose(g["depth"], o["depth"], rtol=2e-6, atol=1e-6)  # the two sides invert the camera matrix independently
    assert np.allclose(g["motion"], o["motion"], atol=2e-3)  # reprojection through two independently inverted matrices (pixels)
    hit = g["normal"][..., 3] == 1.0
    assert np.abs(g["motion"][hit]).max() < 0.05 and (g["motion"][~hit] == 0).all()  # static camera: only the lens / rounding offset remains
    assert (g["normal"][..., 3] == 1.0).mean() > 0.5


def _soup_scene(tmp_path, n_tris, seed, spread=1.0, size=0.15):
    """A triangle soup inside the Cornell box volume, as a Vision scene file: cbox_matte with every shape but the light
    replaced by one OBJ model of n_tris random triangles (n_tris = 0: the light quad is the only geometry)."""
    import json
    text = open(os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json")).read()
    sc = json.loads("\n".join(l for l in text.split("\n") if not l.startswith("//")))
    rng = np.random.default_rng(seed)
    light = [s for s in sc["shapes"] if "emission" in s["param"]]
    sc["shapes"] = light
    if n_tris:
        c = rng.uniform([-spread, 0.0, -spread], [spread, 2.0 * spread, spread], (n_tris, 1, 3))
        v = (c + rng.normal(scale=size, size=(n_tris, 3, 3))).reshape(-1, 3)
        with open(os.path.join(tmp_path, "soup.obj"), "w") as f:
            for p in v:
                f.write(f"v {p[0]:.7g} {p[1]:.7g} {p[2]:.7g}\n")
            for i in range(n_tris):
                f.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")
        sc["shapes"].append({"type": "model", "name": "soup", "param": {"fn": "soup.obj", "material": sc["materials"][0]["name"], "smooth": False,
                                                                       "transform": {"type": "matrix4x4", "param": {"matrix4x4": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]}}}})
    path = os.path.join(tmp_path, f"soup_{n_tris}.json")
    json.dump(sc, open(path, "w"))
    return path


@pytest.mark.parametrize("n_tris", [0, 1, 2, 3, 5, 17, 257, 4099])
def test_traversal_on_ragged_triangle_counts(backend, tmp_path, n_tris):
    """Edge cases of the builder and the quad traversal: scenes smaller than one leaf (<= 4 triangles: the root IS a leaf),
    just above it, and sizes that leave partially filled BVH4 nodes; closest and any-hit results match the oracle's for
    random rays, axis-parallel rays (zero direction components -> infinite slab reciprocals) and rays that start on geometry."""
    from vision_amd.host import HostScene
    from oracle import oracle_py
    hs = HostScene(_soup_scene(str(tmp_path), n_tris, seed=n_tris + 1), width=16, height=16)
    backend.upload_scene(hs)
    info = backend.build_accel()
    assert hs.scene.n_tris == n_tris + 2 and info["depth"] <= info["stack_depth"]
    if hs.scene.n_tris <= 4:
        assert info["n_nodes"] == 0 and info["n_leaves"] == 1
    osc = oracle_py.OracleScene(hs)
    rng = np.random.default_rng(5)
    n = 4096
    org = rng.uniform([-1.2, -0.2, -1.2], [1.2, 2.2, 1.2], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:256] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 256)] * rng.choice([-1.0, 1.0], (256, 1)).astype(np.float32)  # axis-parallel
    tp = np.ctypeslib.as_array(hs.scene.tri_pos, shape=(hs.scene.n_tris,))
    k = rng.integers(0, hs.scene.n_tris, 256)
    org[256:512] = np.stack([np.array(tp[i]["p0"]) * 0.5 + np.array(tp[i]["p1"]) * 0.3 + np.array(tp[i]["p2"]) * 0.2 for i in k]).astype(np.float32)  # on a triangle
    tmax = np.full(n, 3.0e38, np.float32); tmax[::3] = rng.uniform(0.1, 2.0, tmax[::3].shape).astype(np.float32)
    for any_hit in (False, True):
        hg, _ = backend.trace(org, d, tmax, any_hit=any_hit)
        ho = osc.trace(org, d, tmax, any_hit=any_hit)
        assert np.array_equal(hg, ho), (n_tris, any_hit, int((hg != ho).any(1).sum()))


def test_large_scene_build_and_trace(backend, tmp_path):
    """300k-triangle soup: the PLOC + BVH4 build stays within the per-ray stack, and traversal matches the oracle."""
    from vision_amd.host import HostScene
    from oracle import oracle_py
    hs = HostScene(_soup_scene(str(tmp_path), 300000, seed=9, spread=1.0, size=0.01), width=16, height=16)
    backend.upload_scene(hs)
    info = backend.build_accel()
    assert info["depth"] <= info["stack_depth"] and info["n_nodes"] > 30000
    osc = oracle_py.OracleScene(hs)
    rng = np.random.default_rng(6)
    n = 8192
    org = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32); org[:, 1] += 1.0
    d = rng.normal(size=(n, 3)).astype(np.float32)
    tmax = np.full(n, 3.0e38, np.float32)
    hg, _ = backend.trace(org, d, tmax)
    assert np.array_equal(hg, osc.trace(org, d, tmax))
    assert (hg[:, 0] != 0xFFFFFFFF).mean() > 0.5


@pytest.mark.parametrize("w, h, max_depth, min_depth, tiles", [
    (33, 17, 0, 0, None),            # `$for(&bounces, 0, max_depth)` never runs: black film, no rays
    (33, 17, 1, 5, None),            # one bounce, min_depth beyond max_depth
    (1, 1, 4, 0, None),              # a single pixel
    (37, 29, 3, 0, (8, 2, 3)),       # odd size, small tiles, rank 2 of 3
    (37, 29, 3, 0, (64, 5, 7)),      # one tile covers the image: rank 5 owns nothing
    (70, 40, 16, 5, (64, 1, 2)),     # partially covered tiles at the image border
])
def test_render_edge_cases(backend, w, h, max_depth, min_depth, tiles):
    """Film / work-distribution edge cases against the oracle: zero depth, single pixel, sizes that are not tile multiples,
    ranks that own few or no tiles."""
    from vision_amd import _abi
    hs, p, osc, _ = _load(backend, "scenes/cbox/cbox_materials.json", w, h, max_depth=max_depth, min_depth=min_depth)
    t = _abi.Tiles(*tiles) if tiles else None
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, 3, tiles=t)
    backend.render_batch(3, 2, tiles=t)
    img = backend.download_accum()
    cg = backend.counters()
    ref, co = osc.render(p, 0, 5, tiles=t)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for k in ("closest_rays", "shadow_rays", "paths", "surface_hits"):
        assert cg[k] == co[k], (k, cg[k], co[k])
    if max_depth == 0:
        assert cg["closest_rays"] == 0 and (img[..., :3] == 0).all()
    if tiles == (64, 5, 7):
        assert cg["paths"] == 0 and (img == 0).all()


def _render_linear(backend, spp):
    backend.reset_accum()
    done = 0
    while done < spp:
        n = min(128, spp - done)
        backend.render_batch(done, n); done += n
    return backend.download_accum()[..., :3].astype(np.float64)


def test_cbox_prism_agrees_with_the_reference_render(backend):
    """Image-level pin against pixels the REFERENCE produced, on a scene with no stripped asset: res/render_scene/cbox/dispersion-hero.png
    is cbox-prism.json exactly as shipped (LASF9 glass sphere, roughness 0.08, checker back wall, spectrum/hero with 4 wavelengths,
    1024x1024) — decoded to linear radiance by tools/make_golden_refimage.py (one ACES tone map + sRGB, found by fit; 0.9 % of the
    pixels, the clipped image of the lamp in the sphere, are excluded on both sides).  ONE global scale is fitted on the diffuse
    walls; then every region must carry the reference's energy: the interior of the sphere (two refractions through a dispersive
    dielectric, internal reflections, Russian roulette under eta_scale), its rim (grazing Fresnel reflection), the caustic under it
    (BSDF-sampled paths through the glass to the lamp) and the textured back wall.  This is independent of this repo's oracle: host
    encoding, the sphere tessellation, the regenerated rgb -> spectrum table and the whole dielectric chain are on trial.
    The hero render must also show the reference's COLOUR FRINGES (red-blue chroma at the checker edges seen through the glass)
    with the same sign and size; the srgb render of the same file must not."""
    import refimage_util as ru
    z, valid, refb = ru.load("cbox_prism_ref.npz")
    shape = (1024, 1024)
    disc = lambda x, y, r: ((x - 600) ** 2 + (y - 490) ** 2) < r * r
    R = {"red_wall": ru.block_mask(shape, lambda x, y: (x >= 10) & (x < 100) & (y >= 300) & (y < 800)),
         "green_wall": ru.block_mask(shape, lambda x, y: (x >= 930) & (x < 1010) & (y >= 200) & (y < 560)),
         "ceiling": ru.block_mask(shape, lambda x, y: (y >= 20) & (y < 100) & (((x >= 250) & (x < 420)) | ((x >= 600) & (x < 780)))),
         "floor": ru.block_mask(shape, lambda x, y: (x >= 100) & (x < 300) & (y >= 920) & (y < 1000)),
         "back_wall": ru.block_mask(shape, lambda x, y: (x >= 140) & (x < 380) & (y >= 160) & (y < 860)),
         "sphere": ru.block_mask(shape, lambda x, y: disc(x, y, 170)),
         "sphere_rim": ru.block_mask(shape, lambda x, y: disc(x, y, 204) & ~disc(x, y, 175)),
         "caustic": ru.block_mask(shape, lambda x, y: (x >= 520) & (x < 800) & (y >= 920) & (y < 995))}
    walls = R["red_wall"] | R["green_wall"] | R["ceiling"] | R["floor"]
    yy, xx = np.mgrid[330:650, 440:760]
    inner = disc(xx, yy, 160) & valid[330:650, 440:760]
    ref_fr = z["fringe"].astype(np.float64)[inner]
    strong = np.abs(ref_fr) > 3 * ref_fr.std()
    report = {}
    for spectrum in (None, "srgb"):  # as shipped (hero, 4 wavelengths), then the same file under spectrum/srgb
        hs, p, osc, _ = _load(backend, "scenes/cbox/cbox-prism.json", 1024, 1024, **({"spectrum": spectrum} if spectrum else {}))
        lin = _render_linear(backend, 1024)
        mb = ru.block_sums(lin, valid)
        k = refb[walls].sum() / mb[walls].sum()
        ratios = ru.region_ratios(mb * k, refb, R)
        fr = ru.fringe_map(lin * k)[330:650, 440:760][inner]
        agree = float((np.sign(fr[strong]) == np.sign(ref_fr[strong])).mean())
        proj = float((fr[strong] * np.sign(ref_fr[strong])).mean())
        report[spectrum or "hero4"] = (k, {a: [round(float(v), 3) for v in b] for a, b in ratios.items()}, agree, proj)
        print("cbox-prism vs the reference's render:", spectrum or "hero4 (as shipped)", "scale %.4f" % k, report[spectrum or "hero4"][1],
              "fringes: sign agreement %.3f, mean projection %.3f (reference %.3f, %d pixels)" % (agree, proj, float(np.abs(ref_fr[strong]).mean()), int(strong.sum())))
        assert abs(k - 1.0) < 0.06, k  # the picture carries exposure 1: no hidden scale
        luma = lambda r: float(np.dot(r, [0.2126, 0.7152, 0.0722]))
        for name in ("back_wall", "sphere", "sphere_rim", "caustic", "floor", "ceiling"):
            assert abs(luma(ratios[name]) - 1.0) < 0.05, (name, ratios[name])          # energy within the walls' ratio +- 5 %
        for name in ("back_wall", "sphere", "floor") + (("sphere_rim", "caustic") if spectrum is None else ()):
            assert np.abs(ratios[name] - 1.0).max() < 0.05, (name, ratios[name])       # ... and per channel where the light is white
        for name in ("red_wall", "green_wall"):
            assert np.abs(ratios[name] - 1.0).max() < 0.08, (name, ratios[name])       # (the dark channels of the coloured walls sit at 8-bit levels ~40)
    ref_size = float(np.abs(ref_fr[strong]).mean())
    assert strong.sum() > 500 and ref_size > 0.3
    assert report["hero4"][2] > 0.97 and abs(report["hero4"][3] / ref_size - 1.0) < 0.1, report["hero4"]   # the fringes are there, same sign, same size
    assert report["srgb"][3] < 0.15 * ref_size, report["srgb"]                                            # and they are dispersion, not geometry


def test_classroom_texture_detail_agrees_with_the_reference_render(backend):
    """What the textures SHOW, against the reference's own 1024-spp render of classroom (tests/golden/classroom_ref_detail.npz: band-passed
    luminance of the two notice boards, the blackboard and the lectern — image textures on OBJ meshes).  The lighting cannot be compared
    (the environment map is stripped; a procedural sky stands in), local texture detail can: normalised cross-correlation 0.83-0.88 on the
    boards, ~0 for a mirrored or upside-down lookup.  Pins, independently of the twin oracle: OBJ import with flip_uv, uv interpolation,
    the JPEG decoder's row order, instance transforms and the camera (the regions are fixed pixel boxes of the 1280 x 720 frame)."""
    from scipy.ndimage import gaussian_filter
    hs, p, osc, _ = _load(backend, "scenes/classroom/vision_scene.json", 1280, 720)
    lin = _render_linear(backend, 256)
    srgb = lambda x: np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-12), 1 / 2.4) - 0.055)
    lum = srgb(1.0 - np.exp(-lin)) @ np.array([0.2126, 0.7152, 0.0722])
    band = gaussian_filter(lum, 1.0) - gaussian_filter(lum, 6.0)
    z = np.load(os.path.join(ROOT, "tests", "golden", "classroom_ref_detail.npz"))

    def ncc(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    report = {}
    for name, floor in (("left_board", 0.75), ("right_board", 0.75), ("blackboard", 0.7), ("lectern", 0.5)):
        y0, y1, x0, x1 = (int(v) for v in z[name + "_box"])
        mine, ref = band[y0:y1, x0:x1], z[name].astype(np.float64)
        report[name] = (ncc(mine, ref), ncc(mine[:, ::-1], ref), ncc(mine[::-1], ref))
    print("classroom texture detail vs the reference's render (ncc as rendered, mirrored, upside down):", {k: [round(x, 3) for x in v] for k, v in report.items()})
    for name, floor in (("left_board", 0.75), ("right_board", 0.75), ("blackboard", 0.7), ("lectern", 0.5)):
        as_is, mirrored, upside_down = report[name]
        assert as_is > floor and mirrored < 0.4 and upside_down < 0.4, (name, report[name])


def test_glass_of_water_agrees_with_the_reference_render(backend):
    """Second picture of the reference's own: its 1024-spp render of glass-of-water (tests/golden/glass_of_water_ref.npz: decoded
    with the exposure curve 1 - exp(-x) + sRGB that picture carries, clipped sparkles excluded, tools/make_golden_refimage.py).  The
    poured-water mesh (models/Mesh000.obj: the stream, the water in the glass, the splashes) is missing from the checkout, so the blocks
    it covers are excluded — and ONLY those.  Regions: (a) backdrop + far table, (b) the rough-conductor table (measured metal, GGX),
    (c) the three ice cubes (ROUGH dielectric: refraction, internal reflection, max depth 32), (d) the strip under the glass, which
    mirrors the missing water and is only held loosely.  The picture's colour balance is known only by fit (R -11 %), so the regions
    are judged against the backdrop's ratio.
    (Round 2 compared linear sums with sums of 1 - exp(-x) and read a 14-27 % "excess" on the ice cubes: that was the exposure
    curve, which compresses bright regions; decoded properly the cubes sit at 1.00 / 0.94 / 0.92 of the backdrop's ratio.)"""
    import refimage_util as ru
    hs, p, osc, _ = _load(backend, "scenes/glass-of-water/vision_scene.json", 1280, 720)
    lin = _render_linear(backend, 512)
    z, valid, refb = ru.load("glass_of_water_ref.npz")
    shape = (720, 1280)
    blk = lambda r0, r1, c0, c1: ru.block_mask(shape, lambda x, y: (y >= r0 * 16) & (y < r1 * 16) & (x >= c0 * 16) & (x < c1 * 16))
    missing = blk(0, 39, 24, 54)  # the stream, the glass with the water in it, the splashes around its rim
    R = {"ice_cubes": blk(33, 42, 15, 27) | blk(29, 38, 54, 65), "under_glass": blk(39, 45, 24, 54)}
    R["table_metal"] = blk(31, 45, 0, 80) & ~(missing | R["ice_cubes"] | R["under_glass"])
    R["backdrop"] = blk(0, 31, 0, 80) & ~missing
    ratios = ru.region_ratios(ru.block_sums(lin, valid), refb, R)
    rel = {k: v / ratios["backdrop"] for k, v in ratios.items()}
    print("glass-of-water vs the reference's render, linear energy ratio RGB by region:", {k: [round(float(x), 3) for x in v] for k, v in ratios.items()},
          "relative to the backdrop:", {k: [round(float(x), 3) for x in v] for k, v in rel.items()})
    assert np.abs(ratios["backdrop"] - 1.0).max() < 0.13, ratios["backdrop"]
    limits = {"table_metal": 0.06, "ice_cubes": 0.10, "under_glass": 0.12}
    for k, lim in limits.items():
        assert np.abs(rel[k] - 1.0).max() < lim, (k, rel[k])


def test_classroom_sky_through_the_fog_is_attenuated_over_the_world_diameter(backend):
    """integrator.cpp:146-151 on the scene that ships with it: classroom's JSON has global fog (sigma_t = 0.0221 / m, no absorption)
    and the camera inside it.  A sky pixel seen through the window must come out as
        L_fog = L_nofog * exp(-sigma_t * (d_window + world_diameter))  (+ a little in-scattered light),
    world_diameter = 313.4 m for this scene's bounds, d_window = a few metres: a factor of 0.7e-3 ... 1.0e-3 before in-scattering.
    The expected band comes from the scene's numbers, not from the oracle; without the rule the ratio is ~0.9.
    (Why the reference's two classroom PNGs cannot serve as this pin — the factor is global, every photon of the scene comes from
    the environment, and the HDRI's level is unknown — is worked out in DESIGN.md section 2.)"""
    imgs = {}
    for fog in (False, True):
        hs, p, osc, _ = _load(backend, "scenes/classroom/vision_scene.json", 640, 360, mediums=fog)
        if fog:
            wd = hs.scene.lights[hs.scene.env_light].world_diameter
            med = hs.scene.mediums[0]
            sigma_t = (med.sigma_a[0] + med.sigma_s[0]) * med.scale
            assert abs(wd - 313.4375) < 1e-3 and abs(sigma_t - 0.0221) < 1e-6
        backend.reset_accum()
        backend.render_batch(0, 128)
        imgs[fog] = backend.download_accum()[..., :3].astype(np.float64)
    lum = imgs[False] @ np.array([0.212671, 0.715160, 0.072169])
    sky = lum >= np.quantile(lum, 0.985)          # the brightest 1.5 % of the fog-free picture: sky seen through the windows
    ratio = imgs[True][sky].sum() / imgs[False][sky].sum()
    lo, hi = np.exp(-sigma_t * (wd + 15.0)), np.exp(-sigma_t * wd) * 1.6   # up to 15 m to the window; up to +60 % in-scattered light
    print("classroom sky through fog: ratio", ratio, "band", lo, hi)
    assert lo < ratio < hi, (ratio, lo, hi)


class _DevBuf:
    """float32 device buffer through the HIP runtime libvmk.so already loaded (no torch in this process: a second HIP
    runtime — torch bundles its own — does not see the GPU once the first one holds it)."""

    def __init__(self, shape, fill=0.0):
        import ctypes as C
        self._hip = C.CDLL("libamdhip64.so")
        self.shape = shape
        self.nbytes = int(np.prod(shape)) * 4
        p = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(p), C.c_size_t(self.nbytes)) == 0
        self.ptr = p.value
        host = np.full(shape, fill, np.float32)
        assert self._hip.hipMemcpy(C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes), 1) == 0

    def numpy(self):
        import ctypes as C
        out = np.zeros(self.shape, np.float32)
        assert self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(self.nbytes), 2) == 0
        return out

    def free(self):
        import ctypes as C
        self._hip.hipFree(C.c_void_p(self.ptr))


def test_exchange_behind_the_c_abi_with_a_one_rank_communicator(backend):
    """vmk_comm_init / vmk_allreduce_framebuffer / vmk_allgather_framebuffer (RCCL behind the C-ABI) on a 1-rank
    communicator: both forms of the exchange deliver the framebuffer bit for bit into the receive buffer, the exchange of
    batch k overlaps batch k+1 (whose film resolve waits for it), and the asynchronous launch timing returns one time per batch.
    The N = 2 data path is covered on CPU (gloo) in test_oracle_render.py; the 1 -> 8 curve is the driver's run."""
    from vision_amd import _abi
    from vision_amd.backend import Backend
    hs, p, osc, _ = _load(backend, "scenes/cbox/cbox_materials.json", 70, 40)
    fb = _DevBuf((40, 70, 4), 0.0)
    full_a, full_b, full_g = _DevBuf((40, 70, 4), -1.0), _DevBuf((40, 70, 4), -1.0), _DevBuf((40, 70, 4), -1.0)
    backend.set_framebuffer(fb.ptr)
    backend.comm_init(Backend.comm_unique_id(), 0, 1)
    backend.enable_kernel_timing(True)
    backend.reset_accum()
    backend.render_batch(0, 2)                 # batch 0
    backend.allreduce_framebuffer(full_a.ptr)  # its exchange, on the exchange stream
    backend.render_batch(2, 2)                 # batch 1 starts at once; its film resolve waits for the exchange
    backend.allreduce_framebuffer(full_b.ptr)
    backend.allgather_framebuffer(_abi.Tiles(16, 0, 1), full_g.ptr)
    backend.comm_synchronize()
    times = backend.collect_kernel_ms()
    backend.enable_kernel_timing(False)
    assert len(times) == 2 and all(t > 0 for t in times)
    ref2, _ = osc.render(p, 0, 2)
    ref4, _ = osc.render(p, 0, 4)
    a, b, gth, own = full_a.numpy(), full_b.numpy(), full_g.numpy(), fb.numpy()
    assert np.array_equal(a.view(np.uint32), ref2.view(np.uint32))      # the exchange saw batch 0's film, not batch 1's
    assert np.array_equal(b.view(np.uint32), ref4.view(np.uint32))
    assert np.array_equal(gth.view(np.uint32), own.view(np.uint32)) and np.array_equal(own.view(np.uint32), ref4.view(np.uint32))
    with pytest.raises(Exception, match="second"):
        backend.allreduce_framebuffer(fb.ptr)  # in place would double-count on the next batch: refused
    backend.set_framebuffer(None)
    for d in (fb, full_a, full_b, full_g):
        d.free()


def test_config5_bathroom2_4k_sampled_tiles_and_eight_emulated_ranks(backend):
    """BASELINE config 5 (bathroom2 3840x2160, max depth 64) with the declared stand-ins for the assets stripped from the
    reference checkout (vmk_host_options.missing_assets = standin: 10 meshes skipped, WoodPanel.png -> grey, HDRI -> procedural sky).
    (1) the whole 4K frame on the GPU vs the oracle on every 510th 32x32 tile, bit for bit; (2) the full-size property set with
    world = 8 emulated ranks: the eight tile shards are disjoint, cover the image and sum to the single-rank image exactly;
    (3) no rank is pinned to a column set at 3840 px (120 tiles per row = 15 x 8)."""
    from vision_amd import _abi
    hs, p, osc, info = _load(backend, "scenes/bathroom2/vision_scene.json", 3840, 2160, missing_assets="standin", max_depth=64)
    assert p.max_depth == 64 and hs.scene.n_tris > 380000
    assert info["depth"] > info["stack_depth"]  # PLOC's tree needs 94 stack entries: the deep-tree kernel variants (HBM stack overflow) serve it
    assert hs.description.count("stand-in") == 12
    assert backend.self_check() > 4000
    backend.reset_accum(); backend.reset_counters()
    backend.render_batch(0, 2)
    full = backend.download_accum()
    c_full = backend.counters()
    assert np.isfinite(full).all() and (full[..., 3] == 1.0).all() and c_full["paths"] == 3840 * 2160 * 2
    ref, cc = osc.render(p, 0, 2, tiles=_abi.Tiles(32, 77, 510))
    owned = ref[..., 3] != 0.0
    assert owned.sum() == 16 * 32 * 32 and cc["paths"] == owned.sum() * 2
    assert np.array_equal(full[owned].view(np.uint32), ref[owned].view(np.uint32))
    total = np.zeros_like(full)
    cover = np.zeros(full.shape[:2], np.int32)
    paths = 0
    for rank in range(8):
        backend.reset_accum(); backend.reset_counters()
        backend.render_batch(0, 2, tiles=_abi.Tiles(32, rank, 8))
        part = backend.download_accum()
        mine = part[..., 3] != 0.0
        cols = np.nonzero(mine.any(0))[0]
        assert cols.size == 3840, "a rank must not be confined to a column subset"  # t mod 8 would give 480 columns
        assert abs(int(mine.sum()) - 3840 * 2160 // 8) <= 3840 * 32
        cover += mine
        total += part
        paths += backend.counters()["paths"]
    assert (cover == 1).all() and paths == c_full["paths"]
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32))


def test_bench_distributed_path_rehearsal_with_one_rank(tmp_path):
    """bench.py's N > 1 code path on the one GPU a test box has: torch.distributed.run starts ONE rank, --rehearse-dist initialises the RCCL
    process group anyway (so every `if dist:` branch runs: id broadcast, agreement all-reduce, barriers, max-over-ranks timing) and
    --force-exchange creates the C-ABI communicator beside torch's and runs vmk_allreduce_framebuffer after every step.  What a second
    rank would add — another GPU — cannot be rehearsed here; the 2-rank tile arithmetic is covered on the CPU (test_oracle_render.py)."""
    import json, subprocess, sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "c2", "--spp-per-step", "8", "--rehearse-dist", "--force-exchange",
           "--no-cpu-baseline", "--no-other-configs", "--no-replay"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    assert line["config"]["exchange"].startswith("allreduce behind the C-ABI"), line["config"]["exchange"]
    assert line["self_check"].startswith("megakernel == unit kernel")


def test_coffee_maker_agrees_with_the_reference_test_case_picture(backend):
    """Third image-level pin against pixels the REFERENCE produced — and the first on substrate (coated plastic) and mirror materials, three
    quad area lights and twenty OBJ meshes under matrix transforms: res/test_case/coffee/output.png, the expected picture
    of the reference's own test case (tools/auto_test.py runs the scene; nothing there compares the picture).  One mesh of the scene, the
    glass carafe, is not in the checkout; the picture's upper three quarters do not show it.
    Two things had to be found (tools/make_golden_refimage.py): the picture is plain sRGB of the linear accumulation (clipped; 8 % of the
    pixels are excluded on both sides), and it was taken with fov_y 20 — the class default, the file says 25 — and the look_at direction
    mirrored in yaw and pitch, i.e. by an older reading of the same file.  Neither was fitted: with exactly those two changes
    (scenes/coffee/vision_scene_refcam.json) the render aligns with the picture block for block (correlation of log luminance 0.999; 0.89
    with the file's camera as today's reference and this build read it) and then agrees in ABSOLUTE radiance, no scale: backdrop, orange
    body (lit and shadowed side), cap within 5 % per channel.  The black base right above the missing carafe is 16 % brighter (documented,
    bounded); the chrome pipe, a few pixels wide and mirroring the surroundings, within 12 %."""
    import refimage_util as ru
    z, valid, refb = ru.load("coffee_ref.npz")
    shape = (1000, 800)
    box = lambda x0, x1, y0, y1: ru.block_mask(shape, lambda x, y: (x >= x0) & (x < x1) & (y >= y0) & (y < y1))
    R = {"backdrop_left": box(20, 150, 100, 400), "backdrop_right": box(650, 780, 100, 400), "backdrop_top": box(100, 700, 10, 90), "backdrop_low": box(20, 140, 800, 990),
         "body_front": box(420, 560, 250, 640), "body_left": box(260, 330, 250, 560), "cap": box(380, 470, 130, 152), "base": box(300, 520, 690, 740), "pipe": box(160, 186, 520, 740)}
    upper = np.zeros(refb.shape[:2], bool); upper[:740 // 8] = True
    def corr(mb):
        lum = lambda b: np.log(b @ np.array([0.2126, 0.7152, 0.0722]) + 1e-3)
        a, b = lum(mb)[upper], lum(refb)[upper]
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    luma = lambda r: float(np.dot(r, [0.2126, 0.7152, 0.0722]))
    hs, p, osc, _ = _load(backend, "scenes/coffee/vision_scene_refcam.json", 800, 1000, missing_assets="standin")
    assert "Mesh010.obj" in hs.description and hs.description.count("stand-in") == 1  # the carafe, nothing else
    mb = ru.block_sums(_render_linear(backend, 512), valid)
    ratios = ru.region_ratios(mb, refb, R)
    c = corr(mb)
    print("coffee maker vs the reference's test-case picture: block correlation %.4f" % c, {a: [round(float(v), 3) for v in b] for a, b in ratios.items()})
    assert c > 0.995, c
    for name in ("backdrop_left", "backdrop_right", "backdrop_top", "backdrop_low", "body_front", "body_left", "cap"):
        assert np.all(np.abs(ratios[name] - 1.0) < 0.05), (name, ratios[name])  # absolute radiance, per channel
    assert 1.0 < luma(ratios["base"]) < 1.3, ratios["base"]     # lit from below where the carafe is missing
    assert abs(luma(ratios["pipe"]) - 1.0) < 0.12, ratios["pipe"]
    # the same comparison through the camera of the file as shipped must FAIL: the pin sees a camera error
    hs, p, osc, _ = _load(backend, "scenes/coffee/vision_scene.json", 800, 1000, missing_assets="standin")
    assert corr(ru.block_sums(_render_linear(backend, 64), valid)) < 0.95


def test_staircase_agrees_with_the_reference_gallery_picture(backend):
    """Fourth image-level pin, on the one shipped scene that loads with NO stand-in and is made of textures (wood, wallpaper, parquet on OBJ
    meshes): gallery/staircase.png.  Like the coffee picture it is plain sRGB of the linear accumulation taken with fov_y 20 and a mirrored
    pitch (tools/make_golden_refimage.py); unlike it, its yaw had to be FITTED (one parameter), so this pin is the weaker of the two and its
    limits are wider.  Through that camera the render correlates with the picture block for block (0.998; 0.88 through the file's camera) and
    carries its ABSOLUTE radiance, no scale fitted: parquet, both wallpapers, the side of the stairs, the panelling and the chair's seat within
    8 % per channel (measured +3 … +6 %, the blue channel highest — a small systematic excess that is recorded here, not explained), the
    table top within 15 %.  The lamp (a shade around a light, next to clipped pixels) is reported, not asserted."""
    import refimage_util as ru
    z, valid, refb = ru.load("staircase_ref.npz")
    shape = (1280, 720)
    box = lambda x0, x1, y0, y1: ru.block_mask(shape, lambda x, y: (x >= x0) & (x < x1) & (y >= y0) & (y < y1))
    R = {"floor": box(60, 460, 980, 1250), "wallpaper_under_stairs": box(20, 300, 300, 520), "wallpaper_top": box(560, 700, 20, 160), "stair_side": box(560, 700, 480, 900),
         "panel": box(580, 700, 230, 400), "chair_seat": box(280, 420, 750, 775), "table": box(40, 220, 720, 760), "lamp": box(170, 260, 545, 600)}
    def corr(mb):
        lum = lambda b: np.log(b @ np.array([0.2126, 0.7152, 0.0722]) + 1e-3)
        a, b = lum(mb).ravel(), lum(refb).ravel()
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    hs, p, osc, _ = _load(backend, "scenes/staircase/vision_scene_refcam.json", 720, 1280)
    assert "stand-in" not in hs.description
    mb = ru.block_sums(_render_linear(backend, 256), valid)
    ratios = ru.region_ratios(mb, refb, R)
    c = corr(mb)
    print("staircase vs the reference's gallery picture: block correlation %.4f" % c, {a: [round(float(v), 3) for v in b] for a, b in ratios.items()})
    assert c > 0.99, c
    for name in ("floor", "wallpaper_under_stairs", "stair_side", "panel", "chair_seat"):
        assert np.all(np.abs(ratios[name] - 1.0) < 0.08), (name, ratios[name])
    for name in ("wallpaper_top", "table"):
        assert np.all(np.abs(ratios[name] - 1.0) < 0.15), (name, ratios[name])
    hs, p, osc, _ = _load(backend, "scenes/staircase/vision_scene.json", 720, 1280)
    assert corr(ru.block_sums(_render_linear(backend, 32), valid)) < 0.95  # the file's own camera: the pin sees the difference

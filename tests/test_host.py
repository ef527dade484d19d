"""Host logic (no GPU): Vision JSON front-end, per-type defaults, mesh/lights/camera encoding, error behaviour."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from vision_amd import _abi
from vision_amd.host import HostScene, HostError


def _write(tmp_path, name, obj_or_text):
    p = os.path.join(tmp_path, name)
    open(p, "w").write(obj_or_text if isinstance(obj_or_text, str) else json.dumps(obj_or_text))
    return p


def _cbox():
    text = open(os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json")).read()
    return json.loads("\n".join(l for l in text.split("\n") if not l.startswith("//")))


def test_cbox_tables(built):
    hs = HostScene(os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json"))
    s = hs.scene
    assert (s.n_tris, s.n_instances, s.n_materials, s.n_lights) == (36, 8, 8, 1)  # cbox 36 triangles
    assert s.env_light == _abi.INVALID
    l = s.lights[0]
    assert l.type == 0 and l.inst_id == 7 and l.two_sided == 0
    # Light::initialize_slots (light.cpp:19-24): (17,12,4) normalised by its max, factor folded into scale
    assert abs(l.scale - 17.0) < 1e-6 and np.allclose(list(l.color.v), [1.0, 12 / 17, 4 / 17])
    assert l.alias_count == 2 and abs(sum(s.alias_func[l.alias_offset + i] for i in range(2)) - 0.47 * 0.38) < 1e-5
    p = hs.params
    assert (p.width, p.height, p.max_depth, p.min_depth, p.filter_type) == (1024, 1024, 16, 0, 1)
    assert p.tone_mapper == 1 and abs(p.filter_radius[0] - 0.5) < 1e-7
    assert "integrator/pt" in hs.description and "sensor/thin_lens" in hs.description and "light/area" in hs.description


def test_camera_looks_at_target(built):
    """look_at -> yaw/pitch -> c2w (sensor.cpp:73-78,153-162): the centre pixel's ray passes through target_pos."""
    from oracle import oracle_py
    hs = HostScene(os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json"), width=65, height=65)
    p = hs.params_copy()
    p.filter_type = 0; p.filter_radius[0] = p.filter_radius[1] = 0.0  # box filter of zero width: exact pixel centre
    px = np.array([[32, 32, 0]], np.uint32).view(np.float32)
    ray = oracle_py.test_eval_noscene(5, px, 6, params=p)[0]
    assert np.allclose(ray[:3], [0, 1, 6.8], atol=1e-5)
    assert np.allclose(ray[3:], [0, 0, -1], atol=1e-5)
    corner = oracle_py.test_eval_noscene(5, np.array([[0, 0, 0]], np.uint32).view(np.float32), 6, params=p)[0]
    t = np.tan(np.radians(19.5) / 2) * (1 - 1 / 65)
    d = corner[3:] / -corner[5]
    assert np.allclose(d[:2], [-t, t], atol=1e-4)  # raster (0,0) is the top-left: -x, +y


def test_classroom_tables(built):
    hs = HostScene(os.path.join(ROOT, "scenes", "classroom", "vision_scene.json"))
    s = hs.scene
    assert (s.n_tris, s.n_instances, s.n_materials, s.n_lights) == (103832, 79, 45, 1)
    assert s.env_light == 0 and s.lights[0].type == 1 and s.lights[0].res_x == 2048 and s.lights[0].res_y == 1024
    types = [s.materials[i].type for i in range(s.n_materials)]
    assert (types.count(0), types.count(4), types.count(2), types.count(3)) == (36, 5, 3, 1)  # App. C
    assert (hs.params.width, hs.params.height, hs.params.max_depth, hs.params.min_depth) == (1280, 720, 16, 5)
    assert "procedural_sky" in hs.description and "medium/ignored" in hs.description
    # metal/Al: (eta,k) at the three sRGB peak wavelengths (metal.cpp:104-129)
    al = [s.materials[i] for i in range(s.n_materials) if s.materials[i].type == 2][0]
    assert np.allclose(list(al.slot[0].v), [1.2203065, 0.919433, 0.6082539], atol=1e-5)
    assert not (al.flags & 1) and abs(al.slot[2].v[0] - 0.1) < 1e-7  # remapping_roughness false, roughness [0.1,0.1] -> x
    hs2 = HostScene(os.path.join(ROOT, "scenes", "classroom", "vision_scene.json"), width=1920, height=1080)
    assert (hs2.params.width, hs2.params.height) == (1920, 1080)


def test_alias_tables_are_consistent(built):
    """AliasTable::build (alias.h:86-122): reconstructing the distribution from (prob, alias) gives back func/sum."""
    hs = HostScene(os.path.join(ROOT, "scenes", "classroom", "vision_scene.json"))
    s = hs.scene
    l = s.lights[0]
    n = l.alias_count
    prob = np.array([s.alias_prob[l.alias_offset + i] for i in range(n)])
    alias = np.array([s.alias_idx[l.alias_offset + i] for i in range(n)])
    func = np.array([s.alias_func[l.alias_offset + i] for i in range(n)])
    pmf = np.zeros(n)
    np.add.at(pmf, np.arange(n), prob / n)
    np.add.at(pmf, alias, (1 - prob) / n)
    assert np.allclose(pmf, func / func.sum(), atol=2e-6)
    assert abs(l.alias_integral - func.sum() / n) < 1e-6 * max(1.0, func.sum() / n)


def test_json_comments_and_defaults(built, tmp_path):
    sc = _cbox()
    del sc["integrator"]["param"]; del sc["camera"]["param"]["filter"]
    text = "// line comment\n/* block\n comment */\n" + json.dumps(sc).replace('"shapes"', '/*c*/"shapes"', 1)
    hs = HostScene(_write(tmp_path, "s.json", text))
    p = hs.params
    assert (p.max_depth, p.min_depth, p.rr_threshold, p.mis_mode) == (16, 5, 1.0, 0)  # integrator.cpp:59-66 defaults
    assert p.filter_type == 2 and abs(p.filter_radius[0] - 0.5) < 1e-7                # FilterDesc default gaussian, r 0.5
    assert p.env_separate == 0 and abs(p.env_prob - 0.5) < 1e-7 and p.ray_offset_factor == 1.0


@pytest.mark.parametrize("mutate, needle", [
    (lambda s: s["materials"].append({"type": "subsurface", "name": "x", "param": {}}), "material type 'subsurface'"),
    (lambda s: s["integrator"].update(type="rt"), "integrator/rt"),
    (lambda s: s["spectrum"].update(type="hero", param={"dimension": 5}), "dimension 5"),
    (lambda s: s["spectrum"].update(type="rgb"), "spectrum/rgb"),
    (lambda s: s["light_sampler"]["param"]["lights"].append({"type": "ies", "param": {}}), "light/ies"),
    (lambda s: s["camera"]["param"].update(filter={"type": "blackman", "param": {"radius": 1}}), "filter/blackman"),
    (lambda s: s["shapes"].append({"type": "torus", "name": "s", "param": {}}), "shape/torus"),
    (lambda s: s["shapes"][-1]["param"].pop("emission"), "no light"),
])
def test_out_of_scope_features_fail_loudly(built, tmp_path, mutate, needle):
    """Error behaviour: anything outside the hot-path scope is rejected with a message (the reference OC_ERRORs)."""
    sc = _cbox()
    mutate(sc)
    with pytest.raises(HostError) as e:
        HostScene(_write(tmp_path, "bad.json", sc))
    assert needle in str(e.value)


def test_obj_loader_fan_triangulation_and_flip_uv(built, tmp_path):
    obj = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 0.25\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
    _write(tmp_path, "q.obj", obj)
    sc = _cbox()
    sc["shapes"].append({"type": "model", "name": "m", "param": {"fn": "q.obj", "material": "Floor"}})
    hs = HostScene(_write(tmp_path, "s.json", sc))
    s = hs.scene
    assert s.n_tris == 38
    t0, t1 = s.tri_pos[36], s.tri_pos[37]   # quad -> (0,1,2) and (0,2,3): assimp_parser.cpp:284-295
    assert list(t0.p2) == [1, 1, 0] and list(t1.p1) == [1, 1, 0] and list(t1.p2) == [0, 1, 0]
    assert abs(s.tri_attr[36].uv2[1] - 0.75) < 1e-7   # flip_uv default true: v -> 1 - v (model.cpp:30-34)
    assert list(s.tri_attr[36].n0) == [0, 0, 1]


def test_point_spot_lights_and_fitted_filters(built):
    """light/point, light/spot (point.cpp, spot.cpp) and filter/mitchell, filter/sinc (fitted_curve.h) reach the tables."""
    import math
    from vision_amd import _abi
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_lights.json"), width=16, height=16)
    sc = hs.scene
    types = [sc.lights[i].type for i in range(sc.n_lights)]
    assert sorted(types) == [0, 2, 3]  # area (the ceiling quad), point, spot
    spot = [sc.lights[i] for i in range(sc.n_lights) if sc.lights[i].type == 3][0]
    assert abs(spot.cos_angle - math.cos(math.radians(20))) < 1e-6
    # spot.cpp:31 clamps the falloff in DEGREES against the angle in RADIANS before converting: min(2, 0.349) deg
    assert abs(spot.cos_falloff_start - math.cos(math.radians(20) - math.radians(math.radians(20)))) < 1e-6
    assert abs(sum(d * d for d in spot.direction) - 1.0) < 1e-6
    point = [sc.lights[i] for i in range(sc.n_lights) if sc.lights[i].type == 2][0]
    assert list(point.position) == pytest.approx([-0.5, 1.8, 0.0]) and point.scale == pytest.approx(0.2)
    assert hs.params.filter_type == 2  # fitted-curve table
    m = list(hs.params.filter_marginal_func)
    assert all(v >= 0 for v in m) and m[0] > m[-1]  # |mitchell| is largest at the centre
    hs2 = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_sinc.json"), width=16, height=16)
    c = np.array(hs2.params.filter_cond_func, np.float32).reshape(20, 20)
    # windowed sinc with radius 1.5 crosses zero at |x| = 1: the tabulated |f| dips there (column 13 ~ x = 1.0125)
    assert c[0, 13] < c[0, 10] and c[0, 13] < c[0, 16]


def test_power_light_sampler_table(built):
    """lightsampler/power (power.cpp:35-52): alias table over luminance(power()) in light order."""
    import math
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_power.json"), width=16, height=16)
    sc = hs.scene
    assert hs.params.light_sampler == 1 and sc.light_alias_offset != _abi.INVALID
    func = [sc.alias_func[sc.light_alias_offset + i] for i in range(sc.n_lights)]
    for i, f in enumerate(func):
        l = sc.lights[i]
        avg = [l.color.v[k] * l.scale for k in range(3)]
        lum = 0.212671 * avg[0] + 0.715160 * avg[1] + 0.072169 * avg[2]
        if l.type == 2:
            assert f == pytest.approx(4 * math.pi * lum, rel=1e-5)                      # point.cpp:33-35
        elif l.type == 0:
            area = sum(sc.alias_func[l.alias_offset + k] for k in range(l.alias_count))
            assert f == pytest.approx((2 if l.two_sided else 1) * area * math.pi * lum, rel=1e-5)  # area.cpp:87-89
    assert sc.light_alias_integral == pytest.approx(sum(func) / len(func), rel=1e-6)
    uni = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_lights.json"), width=16, height=16)
    assert uni.params.light_sampler == 0 and uni.scene.light_alias_offset == _abi.INVALID


def test_hero_spectrum_tables(built):
    """spectrum/hero (hero.cpp:243-252): CIE tables as 94-sample SPDs, metal / dispersive-glass "spd" slots, uplift table."""
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_hero.json"), width=16, height=16)
    sc = hs.scene
    assert sc.spectrum == _abi.SPECTRUM_HERO and "spectrum/hero" in hs.description
    assert sc.spd_cie_count == 94 and sc.spd_cie_interval == pytest.approx(471.0 / 94)   # spd.cpp:27-33,50-53
    assert sc.cie_y_integral == pytest.approx(106.856895, rel=1e-5)                          # the CIE Y integral (spd.cpp:15-25)
    y = np.array([sc.spd_data[sc.spd_cie[1] + i] for i in range(94)])
    assert y.argmax() == 39 and y[39] == pytest.approx(1.0, abs=1e-3)                       # ybar peaks at 555 nm = sample 39
    mats = {hs.description.split("\n")[1 + i].split()[-1]: sc.materials[i] for i in range(sc.n_materials)}
    glass, metal = mats["ShortBox"], mats["TallBox"]
    assert glass.flags & 4 and glass.slot[1].tex == _abi.SLOT_SPD                           # BK7: dispersive, tabulated ior
    off, n = (np.array(list(glass.slot[1].v[:2]), np.float32).view(np.uint32))
    assert n == 94 and 1.50 < sc.spd_data[off + 93] < sc.spd_data[off] < 1.54               # normal dispersion, SPD::to_list spd.h:41-47
    assert metal.slot[0].tex == _abi.SLOT_SPD and metal.slot[1].tex == _abi.SLOT_SPD
    off, n = (np.array(list(metal.slot[0].v[:2]), np.float32).view(np.uint32))
    assert n == 95 and metal.slot[0].v[2] == pytest.approx(471.0 / 95)
    # srgb scenes carry none of this
    srgb = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_materials.json"), width=16, height=16)
    assert srgb.scene.spectrum == _abi.SPECTRUM_SRGB and not srgb.scene.rgb2spec and srgb.scene.n_spd == 0


def test_hero_uplift_round_trip(built):
    """The regenerated sRGB->spectrum table (csrc/host/rgb2spec_opt.h; Vision's own table is absent from the checkout): the
    uplifted spectrum of an sRGB colour, seen under D65 through the CIE observer, is that colour again.  Evaluated through
    the oracle's decode + Monte-Carlo wavelength sampling (hero.cpp:286-299), i.e. the estimator a render uses."""
    from oracle import oracle_py
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_hero.json"), width=16, height=16)
    osc = oracle_py.OracleScene(hs)
    u = ((np.arange(4096) + 0.5) / 4096).astype(np.float32)
    for rgb in ([0.8, 0.2, 0.1], [0.2, 0.6, 0.9], [0.5, 0.5, 0.5], [0.05, 0.7, 0.05], [1.0, 1.0, 1.0]):
        inp = np.concatenate([np.tile(np.array(rgb, np.float32), (len(u), 1)), u[:, None]], 1)
        o = osc.test_eval(hs.params, 60, inp, 24)
        assert np.all((o[:, 0:3] >= 360) & (o[:, 0:3] <= 830)) and np.all(o[:, 3:6] > 0)
        assert np.all((o[:, 6:9] >= 0) & (o[:, 6:9] <= 1))                                  # albedo spectra stay in [0, 1]
        back = o[:, 21:24].astype(np.float64).mean(0)                                       # illumination = uplift x D65
        # 1.2e-2: the table is fitted on exact 5 nm tables, the renderer integrates against SPD::eval's 471/94 nm spacing (spd.cpp:50-53)
        assert np.allclose(back, rgb, atol=1.2e-2), (rgb, back)
    # unbound colours scale linearly: (3, 2, 0.5) = 6 x (0.5, 1/3, 1/12)
    o6 = osc.test_eval(hs.params, 60, np.array([[3.0, 2.0, 0.5, 0.3]], np.float32), 24)[0]
    o1 = osc.test_eval(hs.params, 60, np.array([[0.5, 1 / 3, 1 / 12, 0.3]], np.float32), 24)[0]
    assert np.allclose(o6[9:12], 6 * o1[6:9], rtol=1e-5)


def test_hero_dimension_4_wavelengths_closed_form_and_round_trip(built, tmp_path):
    """spectrum/hero with "dimension": 4 (cbox-prism.json:692-697; the 4-wavelength megakernel instance / oracle build).
    HeroWavelengthSpectrum::sample_wavelength (hero.cpp:286-299) rotates the hero draw by i / dimension and maps it through
    sample_visible_wavelength (hero.cpp:15-24): lambda_i = 538 - 138.888889 atanh(0.85691062 - 1.82750197 fract(u + i / 4)),
    pdf_i = 0.0039398042 / cosh^2(0.0072 (lambda_i - 538)) — evaluated here in float64, independent of the oracle's code.
    The uplift round trip holds with four samples as with three, and the host takes the dimension from the scene file or the option."""
    import json
    from oracle import oracle_py
    path = os.path.join(ROOT, "scenes/cbox/cbox_hero.json")
    hs = HostScene(path, width=16, height=16, spectrum="hero4")
    assert hs.scene.spectrum == _abi.SPECTRUM_HERO and hs.scene.spectrum_dimension == 4 and "dimension 4" in hs.description
    assert HostScene(path, width=16, height=16).scene.spectrum_dimension == 3
    text = open(path).read()
    sc = json.loads("\n".join(l for l in text.split("\n") if not l.lstrip().startswith("//")))
    sc["spectrum"] = {"type": "hero", "param": {"dimension": 4}}
    import shutil; shutil.copy(os.path.join(ROOT, "scenes/cbox/checker.png"), str(tmp_path))  # texture paths are relative to the scene file
    alt = os.path.join(str(tmp_path), "dim4.json"); json.dump(sc, open(alt, "w"))
    from_file = HostScene(alt, width=16, height=16)
    assert from_file.scene.spectrum_dimension == 4 and from_file.scene.n_tris == hs.scene.n_tris
    osc = oracle_py.OracleScene(hs)
    assert osc.dim == 4
    u = ((np.arange(2048) + 0.5) / 2048).astype(np.float32)
    rgb = np.array([0.2, 0.6, 0.9], np.float32)
    o = osc.test_eval(hs.params, 60, np.concatenate([np.tile(rgb, (len(u), 1)), u[:, None]], 1), 29)
    lam, pdf = o[:, 0:4].astype(np.float64), o[:, 4:8].astype(np.float64)
    up = (u.astype(np.float64)[:, None] + np.arange(4) / 4.0) % 1.0
    lam_ref = 538.0 - 138.888889 * np.arctanh(0.85691062 - 1.82750197 * up)
    # (fract(u + i/4) is taken in float32 by the renderer: 2^-24 in `up` moves lambda by < 3e-5 nm in the body of the domain)
    assert np.abs(lam - lam_ref).max() < 2e-3, np.abs(lam - lam_ref).max()
    assert np.allclose(pdf, 0.0039398042 / np.cosh(0.0072 * (lam - 538.0)) ** 2, rtol=2e-5)
    assert np.all((lam >= 360) & (lam <= 830)) and np.all((o[:, 8:12] >= 0) & (o[:, 8:12] <= 1))
    back = o[:, 26:29].astype(np.float64).mean(0)  # illumination spectrum (uplift x D65) through the CIE observer, 4-sample estimator
    assert np.allclose(back, rgb, atol=1.2e-2), back
    # the three-wavelength build refuses the scene instead of reading four samples as three
    import ctypes as C
    assert not oracle_py.lib(3).orc_scene_create(C.cast(hs.tables, C.c_void_p))


def test_multiply_shader_node(built, tmp_path):
    """render_core/shadernode/math.cpp (BinaryOpNode: every binary node multiplies): constant x constant is folded by the host in
    float32, image x constant travels as a tinted image slot (VMK_SLOT_TINTED); two images, an image scale other than 1 and a multiply
    in a normal slot are refused.  Multiplying by (1, 1, 1) must not change a single bit of the picture."""
    import json
    from oracle import oracle_py
    text = open(os.path.join(ROOT, "scenes", "cbox", "cbox_materials.json")).read()
    base = json.loads("\n".join(l for l in text.split("\n") if not l.lstrip().startswith("//")))
    import shutil; shutil.copy(os.path.join(ROOT, "scenes/cbox/checker.png"), str(tmp_path))
    tex_mat = next(m for m in base["materials"] if isinstance(m["param"].get("color"), dict) and "checker.png" in json.dumps(m["param"]["color"]))
    img_slot = tex_mat["param"]["color"]

    def variant(color, name):
        sc = json.loads(json.dumps(base))
        next(m for m in sc["materials"] if m["name"] == tex_mat["name"])["param"]["color"] = color
        path = os.path.join(str(tmp_path), name + ".json"); json.dump(sc, open(path, "w"))
        return path

    def mul(lhs, rhs, channels="xyz"):
        return {"channels": channels, "node": {"type": "multiply", "param": {"lhs": lhs, "rhs": rhs}}}
    num = lambda v: {"channels": "xyz", "node": {"type": "number", "param": {"value": v}}}
    plain = HostScene(variant(img_slot, "plain"), width=24, height=24)
    ident = HostScene(variant(mul(img_slot, num([1, 1, 1])), "ident"), width=24, height=24)
    tint = HostScene(variant(mul(num([0.9, 1.0, 0.5]), img_slot), "tint"), width=24, height=24)   # constant on the left works too
    mid = [m["name"] for m in base["materials"]].index(tex_mat["name"])
    s_plain, s_ident, s_tint = (h.scene.materials[mid].slot[0] for h in (plain, ident, tint))
    assert not (s_plain.tex & _abi.SLOT_TINTED) and (s_ident.tex & _abi.SLOT_TINTED) and (s_tint.tex & _abi.SLOT_TINTED)
    assert (s_ident.tex & 0x3FFFFF) == (s_plain.tex & 0x3FFFFF) and list(s_tint.v) == pytest.approx([0.9, 1.0, 0.5])
    img = {k: oracle_py.OracleScene(h).render(h.params_copy(), 0, 2)[0] for k, h in (("plain", plain), ("ident", ident), ("tint", tint))}
    assert np.array_equal(img["plain"].view(np.uint32), img["ident"].view(np.uint32))       # x * 1.0 is exact
    assert not np.array_equal(img["plain"], img["tint"]) and img["tint"][..., 2].sum() < img["plain"][..., 2].sum()
    # constant x constant folds to the product (float32)
    const = HostScene(variant(mul(num([0.5, 0.25, 2.0]), num([0.5, 2.0, 0.25]), "zyx"), "const"), width=24, height=24)
    sc = const.scene.materials[mid].slot[0]
    assert sc.tex == _abi.INVALID and list(sc.v) == [0.5, 0.5, 0.25]                                # (z, y, x) of (0.25, 0.5, 0.5)
    for bad, msg in ((mul(img_slot, img_slot), "two images"),
                     (mul({"channels": "xyz", "node": {"type": "image", "param": {"fn": "checker.png", "color_space": "srgb", "scale": 2.0}}}, num([1, 1, 1])), "scale other than 1"),
                     (mul(mul(img_slot, num([1, 1, 1])), num([1, 1, 1])), "nested multiply")):
        with pytest.raises(HostError, match=msg):
            HostScene(variant(bad, "bad"), width=24, height=24)
    # the reference's playground scene carries one such node and loads as shipped
    pg = HostScene(os.path.join(ROOT, "scenes/playground/vision_scene.json"), width=32, height=32)
    assert sum(1 for i in range(pg.scene.n_materials) for j in range(18) if pg.scene.materials[i].slot[j].tex not in (_abi.INVALID, 0xFFFFFFFD) and pg.scene.materials[i].slot[j].tex & _abi.SLOT_TINTED) == 1


def test_bathroom2_loads_with_declared_standins(built):
    """BASELINE config 5's scene: the reference checkout lacks 10 meshes, WoodPanel.png and the HDRI.  Without the option the
    loader fails on the first missing asset; with vmk_host_options.missing_assets = standin every substitution is listed."""
    path = os.path.join(ROOT, "scenes/bathroom2/vision_scene.json")
    with pytest.raises(HostError, match="file missing"):
        HostScene(path, width=64, height=36)
    hs = HostScene(path, width=64, height=36, missing_assets="standin")
    lines = [l for l in hs.description.split("\n") if "stand-in" in l]
    assert sum("model_skipped" in l for l in lines) == 10 and any("WoodPanel.png" in l and "constant_grey" in l for l in lines)
    assert any("spruit_sunrise_2k.hdr" in l and "procedural_sky" in l for l in lines)
    assert hs.scene.n_tris == 383795 and hs.scene.n_lights == 2 and hs.params.max_depth == 64


def test_malformed_assets_fail_instead_of_hanging(built, tmp_path):
    """A face record strtol cannot advance over, and a truncated / zero-run RLE scanline in a .hdr, used to spin forever."""
    import json
    text = open(os.path.join(ROOT, "scenes", "cbox", "cbox_matte.json")).read()
    sc = json.loads("\n".join(l for l in text.split("\n") if not l.startswith("//")))
    with open(os.path.join(tmp_path, "bad.obj"), "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x3\n")
    with open(os.path.join(tmp_path, "ok.obj"), "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 # trailing comment\n")
    ident = {"type": "matrix4x4", "param": {"matrix4x4": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]}}
    for fn, ok in (("ok.obj", True), ("bad.obj", False)):
        sc2 = dict(sc); sc2["shapes"] = list(sc["shapes"]) + [{"type": "model", "name": "m", "param": {"fn": fn, "material": sc["materials"][0]["name"], "transform": ident}}]
        path = os.path.join(tmp_path, fn + ".json"); json.dump(sc2, open(path, "w"))
        if ok:
            assert HostScene(path, width=16, height=16).scene.n_tris == 37
        else:
            with pytest.raises(HostError, match="malformed face"):
                HostScene(path, width=16, height=16)
    # truncated RLE .hdr as an environment map: refused by the native decoder, then reported as undecodable (no hang, no huge allocation)
    with open(os.path.join(tmp_path, "env.hdr"), "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 8 +X 16\n" + bytes([2, 2, 0, 16, 0]))
    with open(os.path.join(tmp_path, "neg.hdr"), "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y -8 +X 16\n")
    for fn in ("env.hdr", "neg.hdr"):
        sc3 = dict(sc); sc3["light_sampler"] = {"type": "uniform", "param": {"lights": [{"type": "spherical", "param": {"color": {"fn": fn, "color_space": "linear"}, "scale": 1}}]}}
        path = os.path.join(tmp_path, fn + ".json"); json.dump(sc3, open(path, "w"))
        with pytest.raises(HostError, match="nor decodable"):
            HostScene(path, width=16, height=16, procedural_env=False)


def test_sphere_tessellation_and_normal_slot(built, tmp_path):
    """shape/sphere (sphere.cpp:20-88): 2 x sub_div^2 x 2 - ... triangles in the reference's vertex / triangle order, unit normals;
    a material's "normal" slot sets VMK_MATF_HAS_NORMAL; mix / add accept ONE principled_bsdf child (LobeSet::flatten)."""
    sc = _cbox()
    ident = {"type": "matrix4x4", "param": {"matrix4x4": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 1, 0, 1]]}}
    sc["shapes"].append({"type": "sphere", "name": "ball", "param": {"radius": 0.25, "sub_div": 8, "material": sc["materials"][0]["name"], "transform": ident}})
    sc["materials"][0]["param"]["normal"] = [0.1, 0.2, 0.97]
    hs = HostScene(_write(tmp_path, "sph.json", sc), width=16, height=16)
    theta, phi = 8, 16
    assert hs.scene.n_tris == 36 + phi * 2 + (theta - 2) * phi * 2
    tp = np.ctypeslib.as_array(hs.scene.tri_pos, shape=(hs.scene.n_tris,))
    ball = tp[tp["inst"] == hs.scene.n_instances - 1]
    pts = np.concatenate([np.array(ball["p0"]), np.array(ball["p1"]), np.array(ball["p2"])])
    assert np.allclose(np.linalg.norm(pts - np.array([0, 1, 0]), axis=1), 0.25, atol=1e-6)
    m0 = hs.scene.materials[0]
    assert m0.flags & 8 and np.allclose(list(m0.normal.v), [0.1, 0.2, 0.97])
    # mix with one principled child is in scope, with two it is not
    pr = {"type": "principled_bsdf", "name": "p", "param": {"color": [0.5, 0.5, 0.5]}}
    sc["materials"].append({"type": "mix", "name": "mx", "param": {"mat0": pr, "mat1": {"type": "diffuse", "name": "d", "param": {}}, "frac": 0.4}})
    HostScene(_write(tmp_path, "mx.json", sc), width=16, height=16)
    sc["materials"][-1]["param"]["mat1"] = dict(pr, name="p2")
    with pytest.raises(HostError, match="two lobe-set"):
        HostScene(_write(tmp_path, "mx2.json", sc), width=16, height=16)


def test_native_image_decoders_match_pillow(built):
    """csrc/host/image_codec.h (own PNG / baseline-JPEG decoders, IJG arithmetic) against Pillow on every image file the scenes
    ship: byte-identical texture tables whichever side decodes, so goldens and GPU parity do not depend on the decoder."""
    import hashlib
    for scene, kw in (("scenes/classroom/vision_scene.json", {}), ("scenes/bathroom2/vision_scene.json", {"missing_assets": "standin"}),
                      ("scenes/cbox/cbox_materials.json", {})):
        digests = []
        for decode in ("native", "pillow"):
            hs = HostScene(os.path.join(ROOT, scene), width=32, height=32, decode=decode, **kw)
            assert (decode == "pillow") == bool(hs.image_paths)
            tex = C.string_at(hs.scene.tex_data, hs.scene.tex_bytes)
            digests.append((hs.scene.n_textures, hs.scene.tex_bytes, hashlib.sha256(tex).hexdigest()))
            hs.close()
        assert digests[0] == digests[1] and digests[0][0] >= 1, (scene, digests)


def test_native_decoders_refuse_what_they_do_not_decode(built, tmp_path):
    from PIL import Image
    sc = _cbox()
    rng = np.random.default_rng(1)
    img = Image.fromarray(rng.integers(0, 255, (24, 40, 3), dtype=np.uint8))
    img.save(os.path.join(tmp_path, "prog.jpg"), progressive=True)
    img.save(os.path.join(tmp_path, "base.jpg"), quality=90, subsampling=1)   # 4:2:2 -> the h2v1 triangle upsampler
    img.convert("P").save(os.path.join(tmp_path, "pal.png"))
    Image.fromarray(rng.integers(0, 255, (24, 40), dtype=np.uint8)).save(os.path.join(tmp_path, "gray.jpg"))
    for fn, native in (("base.jpg", True), ("pal.png", True), ("gray.jpg", True), ("prog.jpg", False)):
        sc["materials"][0]["param"]["color"] = {"fn": fn, "color_space": "srgb"}
        path = _write(tmp_path, fn + ".json", sc)
        hs = HostScene(path, width=16, height=16)          # falls back to Pillow + vmk_host_register_image where needed
        assert bool(hs.image_paths) == (not native)
        ref = HostScene(path, width=16, height=16, decode="pillow")
        assert C.string_at(hs.scene.tex_data, hs.scene.tex_bytes) == C.string_at(ref.scene.tex_data, ref.scene.tex_bytes), fn


# ---- OpenEXR / save path in the C++ host (image_pool.cpp:13-35, pipeline.cpp:190-204,337-354; csrc/host/exr.h, image_codec.h) ----
def test_exr_decoder_on_the_reference_s_own_files(built):
    """Two EXR files the reference ships (tools/make_golden_exr.py): res/sky.exr as it is (RGBA half, ZIP) and two PIZ blocks of
    cbox/TungstenRender.exr.  Pins: (a) this decoder's committed statistics (regression); (b) an independent witness for PIZ — the PNG the
    reference keeps next to that EXR is the same render through a display curve, so decoded^(1/2.2) must track it pixel by pixel;
    (c) plausibility of the sky: finite, non-negative, alpha 1, brighter above the horizon than below."""
    import json
    from vision_amd.host import load_image
    G = os.path.join(ROOT, "tests", "golden")
    exp = json.load(open(os.path.join(G, "exr_expected.json")))
    for fn, e in exp.items():
        a = load_image(os.path.join(G, fn))
        assert a.dtype == np.float32 and list(a.shape) == e["shape"] and np.isfinite(a).all()
        assert np.allclose(a.astype(np.float64).mean((0, 1)), e["mean"], rtol=1e-12) and float(a.max()) == e["max"] and float(a.min()) == e["min"]
        for y, x, v in e["probe"]:
            assert a[y, x].astype(np.float64).tolist() == v
    sky = load_image(os.path.join(G, "exr_sky_zip_half.exr"))
    assert (sky >= 0).all() and (sky[..., 3] == 1).all() and sky[:100, :, :3].mean() > 3 * sky[200:, :, :3].mean()
    box = load_image(os.path.join(G, "exr_cbox_piz_half.exr")).astype(np.float64)
    png = np.load(os.path.join(G, "exr_cbox_rows.npy")).astype(np.float64) / 255.0
    enc = np.clip(box, 0, 1) ** (1 / 2.2)
    assert np.corrcoef(enc.ravel(), png.ravel())[0, 1] > 0.99 and np.abs(enc - png).mean() < 0.06


def test_exr_png_hdr_writers_round_trip(built, tmp_path):
    """vmk_host_save_image: .exr (float32, ZIP) decodes back to the same bits; .png is the 8-bit picture Pillow reads back; .hdr
    within RGBE precision; the deflate encoder's streams pass both this repo's inflate and Pillow's zlib."""
    from PIL import Image
    from vision_amd.host import load_image, save_image, final_picture_mode, HostError
    rng = np.random.default_rng(5)
    img = np.zeros((37, 53, 4), np.float32)
    img[..., :3] = rng.random((37, 53, 3), dtype=np.float32) ** 3 * 4.0
    img[5:20, 7:30, :3] = 0.25          # flat areas: runs for the LZ77 matcher
    img[..., 3] = 1.0
    p = save_image(os.path.join(tmp_path, "a.exr"), img)
    back = load_image(p)
    assert back.shape == (37, 53, 3) and np.array_equal(back.view(np.uint32), img[..., :3].copy().view(np.uint32))
    p = save_image(os.path.join(tmp_path, "a.png"), img)
    want = (np.clip(img[..., :3], 0, 1) * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(np.asarray(Image.open(p)), want) and np.array_equal(load_image(p), want)
    p = save_image(os.path.join(tmp_path, "a.hdr"), img)
    hdr = load_image(p)[..., :3]
    assert np.abs(hdr - img[..., :3]).max() <= img[..., :3].max() / 128.0
    assert final_picture_mode("x.png") == 1 and final_picture_mode("dispersion-hero-2000.exr") == 2 and final_picture_mode("a.hdr") == 2
    with pytest.raises(HostError, match="no encoder"):
        save_image(os.path.join(tmp_path, "a.bmp"), img)
    with pytest.raises(HostError, match="cannot open|no decoder"):
        load_image(os.path.join(tmp_path, "missing.exr"))


def test_truncated_and_hostile_image_files_are_errors_not_hangs(built, tmp_path):
    """A truncated PNG / EXR must fail with a message: the inflate stops as soon as it would have to invent input bits and never grows
    its output beyond what the container declares (a truncated stream used to decode zeros for ever)."""
    from PIL import Image
    from vision_amd.host import load_image, HostError
    rng = np.random.default_rng(2)
    Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(os.path.join(tmp_path, "ok.png"))
    data = open(os.path.join(tmp_path, "ok.png"), "rb").read()
    # keep the chunk structure valid but cut the IDAT payload short (length + CRC are not checked by the decoder)
    at = data.index(b"IDAT")
    ln = int.from_bytes(data[at - 4:at], "big")
    cut = data[:at - 4] + (ln // 2).to_bytes(4, "big") + data[at:at + 4 + ln // 2] + b"\0\0\0\0" + data[at + 4 + ln + 4:]
    open(os.path.join(tmp_path, "cut.png"), "wb").write(cut)
    with pytest.raises(HostError, match="truncated|too short|deflate"):
        load_image(os.path.join(tmp_path, "cut.png"))
    exr = open(os.path.join(ROOT, "tests", "golden", "exr_sky_zip_half.exr"), "rb").read()
    open(os.path.join(tmp_path, "cut.exr"), "wb").write(exr[:len(exr) // 2])
    with pytest.raises(HostError, match="truncated|beyond"):
        load_image(os.path.join(tmp_path, "cut.exr"))
    open(os.path.join(tmp_path, "junk.exr"), "wb").write(b"\x76\x2f\x31\x01" + bytes(64))
    with pytest.raises(HostError):
        load_image(os.path.join(tmp_path, "junk.exr"))


def test_exr_environment_map_loads_when_present(built, tmp_path):
    """A scene whose spherical light names an .exr file gets that file (not the procedural stand-in) when it exists."""
    import shutil
    sc = _cbox()
    shutil.copy(os.path.join(ROOT, "tests", "golden", "exr_sky_zip_half.exr"), os.path.join(tmp_path, "sky.exr"))
    sc["light_sampler"]["param"]["lights"] = [{"type": "spherical", "param": {"color": {"fn": "sky.exr", "color_space": "linear"}, "scale": 1.0,
                                                                               "o2w": {"type": "Euler", "param": {"yaw": 0}}}}]
    hs = HostScene(_write(tmp_path, "exr_env.json", sc), width=16, height=16)
    assert "stand-in" not in hs.description
    lights = [hs.scene.lights[i] for i in range(hs.scene.n_lights)]
    env = [l for l in lights if l.type == 1][0]
    assert (env.res_x, env.res_y) == (512, 256)

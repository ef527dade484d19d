"""Oracle render regression + multi-rank tile sharding on CPU (gloo, world_size 2) — no GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from vision_amd import _abi
from vision_amd.host import HostScene
from oracle import oracle_py

CASES = [("cbox_matte", "scenes/cbox/cbox_matte.json", 32, 32, 8), ("cbox_materials", "scenes/cbox/cbox_materials.json", 32, 32, 4),
         ("classroom", "scenes/classroom/vision_scene.json", 48, 27, 2),
         # participating media honoured (vmk_host_options.mediums = 1): global fog + a material-less smoke volume; classroom's own fog
         ("cbox_media", "scenes/cbox/cbox_media.json", 32, 32, 4), ("classroom_fog", "scenes/classroom/vision_scene.json", 48, 27, 2),
         # point + spot lights, mitchell / Lanczos-sinc pixel filters (SURVEY 8f rank 2 without the spectral part)
         ("cbox_lights", "scenes/cbox/cbox_lights.json", 32, 32, 4), ("cbox_sinc", "scenes/cbox/cbox_sinc.json", 32, 32, 4),
         # BASELINE config 4 in srgb mode: rough / smooth dielectrics + conductors, max depth 32 (divergence stress)
         ("glass_of_water", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),
         ("cbox_power", "scenes/cbox/cbox_power.json", 32, 32, 4),  # lightsampler/power over area + point + spot
         ("cbox_sheen", "scenes/cbox/cbox_sheen.json", 32, 32, 4),  # principled_bsdf with its sheen (LTC) layer
         ("cbox_extra", "scenes/cbox/cbox_extra.json", 32, 32, 4),
         # Material::compute_shading_frame with "normal" slots, mix / add with a principled_bsdf child (LobeSet::flatten), shape/sphere
         ("cbox_normal", "scenes/cbox/cbox_normal.json", 32, 32, 4),
         # the reference's playground scene as shipped: a "multiply" shader node (checker.jpg x a constant), a normal map, mix, principled
         ("playground", "scenes/playground/vision_scene.json", 48, 48, 2),
         # a complete shipped scene (no stripped asset): 19 meshes, 13 lights, 3 JPG textures, 21 materials
         ("staircase2", "scenes/staircase2/vision_scene.json", 48, 27, 2),
         # the reference's default Cornell scene as shipped: spot + point + spherical + projector (Painting3.jpg) + area light in one scene
         ("cbox_vision_scene", "scenes/cbox/vision_scene.json", 48, 48, 2),  # material/metallic and material/add
         # spectrum/hero: all material families incl. dispersive BK7 + measured Cu; diffuse only; media; point + spot
         # lights; config 4 as worded ("spectral glass"); classroom with its environment map and image textures
         ("cbox_hero", "scenes/cbox/cbox_hero.json", 32, 32, 4), ("cbox_hero_matte", "scenes/cbox/cbox_hero_matte.json", 32, 32, 4),
         ("cbox_hero_media", "scenes/cbox/cbox_hero_media.json", 32, 32, 4), ("cbox_hero_lights", "scenes/cbox/cbox_hero_lights.json", 32, 32, 4),
         ("glass_of_water_hero", "scenes/glass-of-water/vision_scene.json", 48, 48, 2), ("classroom_hero", "scenes/classroom/vision_scene.json", 48, 27, 2),
         # spectrum/hero with "dimension": 4 (cbox-prism.json:692-697): four wavelengths per path — the vmk_hero4.hip instance / the ORC_SPEC_DIM = 4 oracle build
         ("cbox_hero4", "scenes/cbox/cbox_hero.json", 32, 32, 4), ("cbox_hero4_matte", "scenes/cbox/cbox_hero_matte.json", 32, 32, 4),
         ("cbox_hero4_media", "scenes/cbox/cbox_hero_media.json", 32, 32, 4), ("glass_of_water_hero4", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),
         ("cbox_prism_hero4", "scenes/cbox/cbox-prism.json", 48, 48, 2)]  # the reference's own dimension-4 scene, as shipped
MEDIA = {"cbox_media", "classroom_fog", "cbox_hero_media", "cbox_hero4_media"}
SPECTRUM = {"glass_of_water_hero": "hero", "classroom_hero": "hero",  # shipped scenes with the spectrum forced to hero (vmk_host_options.spectrum)
            "cbox_hero4": "hero4", "cbox_hero4_matte": "hero4", "cbox_hero4_media": "hero4", "glass_of_water_hero4": "hero4"}  # ... to hero with four wavelengths


@pytest.mark.parametrize("name, path, w, h, spp", CASES)
def test_oracle_matches_committed_golden(built, name, path, w, h, spp):
    hs = HostScene(os.path.join(ROOT, path), width=w, height=h, mediums=name in MEDIA, spectrum=SPECTRUM.get(name))
    assert bool(hs.params.process_mediums) == (name in MEDIA)
    assert hs.scene.spectrum_dimension == (4 if "hero4" in name else 3)
    assert (hs.scene.spectrum == _abi.SPECTRUM_HERO) == ("hero" in name)
    img, cnt = oracle_py.OracleScene(hs).render(hs.params_copy(), 0, spp)
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}_{w}x{h}x{spp}.npy"))
    assert np.array_equal(img.view(np.uint32), gold.view(np.uint32))  # deterministic: bit-exact across threads/runs
    assert np.isfinite(img).all() and cnt["paths"] == w * h * spp
    if name in MEDIA:  # scattering events inside the medium also cast a shadow ray (integrator.cpp:241-243,271-279)
        assert cnt["shadow_rays"] > cnt["surface_hits"]
        plain = HostScene(os.path.join(ROOT, path), width=w, height=h)
        assert not plain.params.process_mediums and plain.scene.n_mediums == 0  # default: the non-fog variant
    else:
        assert cnt["shadow_rays"] == cnt["surface_hits"]  # one shadow ray per shaded vertex


def test_batch_split_and_tile_sharding_are_exact(built):
    """Frames accumulate by the running-mean recurrence, so 8 frames == 5 + 3 frames; tiles are disjoint, so the
    sum over ranks equals the single-rank image bit for bit (x + 0 is exact) — SURVEY.md §8e."""
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_materials.json"), width=40, height=24)
    osc = oracle_py.OracleScene(hs)
    p = hs.params_copy()
    full, _ = osc.render(p, 0, 8)
    part, _ = osc.render(p, 0, 5)
    part, _ = osc.render(p, 5, 3, accum=part)
    assert np.array_equal(full.view(np.uint32), part.view(np.uint32))
    total = np.zeros_like(full)
    owned = np.zeros(full.shape[:2], np.int32)
    for rank in range(3):
        img, _ = osc.render(p, 0, 8, tiles=_abi.Tiles(8, rank, 3))
        owned += (img[..., 3] != 0)
        total += img
    assert (owned == 1).all()
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32))


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VMK_ROOT"])
from vision_amd import _abi
from vision_amd.host import HostScene
from oracle import oracle_py
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
hs = HostScene(os.path.join(os.environ["VMK_ROOT"], "scenes/cbox/cbox_matte.json"), width=48, height=32)
osc = oracle_py.OracleScene(hs)
p = hs.params_copy()
img, _ = osc.render(p, 0, 4, tiles=_abi.Tiles(16, rank, world), threads=2)
t = torch.from_numpy(img)
dist.all_reduce(t, op=dist.ReduceOp.SUM)           # the one collective of the path: framebuffer all-reduce
if rank == 0:
    full, _ = osc.render(p, 0, 4, threads=2)
    assert np.array_equal(t.numpy().view(np.uint32), full.view(np.uint32)), "all-reduced image != single-rank image"
    print("DIST_OK")
dist.destroy_process_group()
'''


def test_two_rank_allreduce_reassembles_the_image(built, tmp_path):
    """N>1 path on CPU: one process per rank, gloo all-reduce of the float4 framebuffer (RCCL on the GPU box)."""
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, VMK_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), script],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DIST_OK" in out.stdout


def test_oracle_aov_planes_are_consistent(built):
    """The G-buffer restatement (frame_buffer.cpp:156-219): unit shading normals on hits, positive linear depth, the
    emitter plane is non-zero exactly on the light, and a Lambert surface's albedo is its colour slot."""
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_matte.json"), width=48, height=48)
    a = oracle_py.OracleScene(hs).render_aov(hs.params_copy(), frame=0)
    hit = a["normal"][..., 3] == 1.0
    assert hit.mean() > 0.9
    assert np.allclose(np.linalg.norm(a["normal"][hit][:, :3], axis=1), 1.0, atol=1e-5)
    assert (a["depth"][hit] > 0).all() and (a["depth"][~hit] == 0).all()
    assert np.abs(a["motion"][hit]).max() < 0.05 and (a["motion"][~hit] == 0).all()  # static pinhole camera: reprojection lands on p_film
    lit = a["emission"][..., :3].sum(-1) > 0
    assert 0 < lit.sum() < hit.sum() * 0.2
    # every material of cbox_matte is diffuse: albedo in [0, 1], and the distinct albedo values are the scene's colours
    alb = a["albedo"][hit][:, :3]
    assert alb.min() >= 0.0 and alb.max() <= 1.0 and len(np.unique(alb.round(5), axis=0)) <= hs.scene.n_materials + 1


def test_config0_cbox_256x256_16spp_on_cpu(built):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): cbox 256x256 at 16 spp on the CPU restatement.
    The result is deterministic down to the bit (own elementary functions, no contraction), so it is pinned by a digest."""
    import hashlib
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_matte.json"), width=256, height=256)
    img, cnt = oracle_py.OracleScene(hs).render(hs.params_copy(), 0, 16)
    assert cnt["paths"] == 256 * 256 * 16 and (cnt["closest_rays"], cnt["shadow_rays"]) == (3266952, 2837842)
    assert np.isfinite(img).all() and abs(float(img[..., :3].mean()) - 0.12030959) < 1e-6
    assert hashlib.sha1(img.tobytes()).hexdigest() == "289c353a54e7b19d6027543dc0abaec98e5f05a3"


@pytest.mark.parametrize("scene", ["scenes/cbox/cbox_matte.json", "scenes/cbox/cbox_lights.json"])
def test_hero_spectrum_converges_to_the_srgb_image(built, scene):
    """Physical pin of the hero path (hero.cpp): colours uplifted to spectra, transported at three sampled wavelengths and
    integrated back through the CIE observer give the image the RGB transport gives — equal for direct light, and equal up
    to metamerism (a few % in the weakest channel) after diffuse interreflection.  512 spp of the same 16x16 film."""
    mean = {}
    for sp in ("srgb", "hero", "hero4"):
        hs = HostScene(os.path.join(ROOT, scene), width=16, height=16, spectrum=sp)
        assert hs.scene.spectrum == (_abi.SPECTRUM_SRGB if sp == "srgb" else _abi.SPECTRUM_HERO)
        img, _ = oracle_py.OracleScene(hs).render(hs.params_copy(), 0, 512)
        assert np.isfinite(img).all()
        mean[sp] = img[..., :3].astype(np.float64).reshape(-1, 3).mean(0)
    for sp in ("hero", "hero4"):  # three and four wavelengths per path estimate the same integral
        ratio = mean[sp] / mean["srgb"]
        assert abs(ratio[0] - 1) < 0.015 and abs(ratio[1] - 1) < 0.015 and abs(ratio[2] - 1) < 0.05, (sp, ratio)


def test_tile_ownership_is_a_latin_square_of_tiles(built):
    """include/vmk.h vmk_tiles: owner(tx, ty) = (tx + skew * ty) mod world.  At 3840 px (120 tiles per row, a multiple of 8)
    `t mod world` would pin every rank to fixed 32-px column stripes; with the skewed lattice every rank owns exactly one tile
    in each row and each column of any world x world block, for every world size the bench uses.  The product's skew
    (vmk_tile_skew, callable without a GPU) and the oracle's restatement agree pixel for pixel."""
    import ctypes as C
    from vision_amd.backend import lib_path
    L = C.CDLL(lib_path())
    L.vmk_tile_skew.argtypes = [C.c_uint32]; L.vmk_tile_skew.restype = C.c_uint32
    tiles_x, tiles_y = 3840 // 32, (2160 + 31) // 32
    for world in (2, 3, 4, 6, 8, 16):
        skew = L.vmk_tile_skew(world)
        assert np.gcd(skew, world) == 1 and skew % 2 == 1
        ty, tx = np.mgrid[0:tiles_y, 0:tiles_x]
        owner = (tx + skew * ty) % world
        counts = np.bincount(owner.ravel(), minlength=world)
        assert counts.max() - counts.min() <= tiles_y  # balanced to within one tile per row
        for r in range(world):
            cols = [set(np.nonzero(owner[y] == r)[0] % world) for y in range(world)]
            assert len(set(map(frozenset, cols))) == world, "a rank owns the same columns in consecutive tile rows"
        blk = owner[:world, :world]
        for r in range(world):  # Latin square: once per row and once per column of the block
            assert ((blk == r).sum(0) == 1).all() and ((blk == r).sum(1) == 1).all()
    # the oracle's ownership (used by every sharding parity test) is the same function
    hs = HostScene(os.path.join(ROOT, "scenes/cbox/cbox_matte.json"), width=96, height=64)
    osc = oracle_py.OracleScene(hs)
    for world in (2, 3, 8):
        skew = L.vmk_tile_skew(world)
        seen = np.zeros((64, 96), np.int32)
        for rank in range(world):
            img, _ = osc.render(hs.params_copy(), 0, 1, tiles=_abi.Tiles(8, rank, world))
            mine = img[..., 3] != 0
            py, px = np.mgrid[0:64, 0:96]
            assert np.array_equal(mine, ((px // 8 + skew * (py // 8)) % world) == rank)
            seen += mine
        assert (seen == 1).all()

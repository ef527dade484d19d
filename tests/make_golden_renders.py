#!/usr/bin/env python3
"""Golden radiance fixtures: linear accumulation buffers rendered by the CPU oracle (tests/golden/*.npy).

SURVEY.md §8c fixture (iii): no linear-radiance golden exists in the reference tree, so the GPU parity fixtures are
the CPU restatement's output on fixed (scene, resolution, frame range).   python tests/make_golden_renders.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.host import HostScene
from oracle import oracle_py

CASES = [("cbox_matte", "scenes/cbox/cbox_matte.json", 32, 32, 8), ("cbox_materials", "scenes/cbox/cbox_materials.json", 32, 32, 4),
         ("classroom", "scenes/classroom/vision_scene.json", 48, 27, 2),
         ("cbox_media", "scenes/cbox/cbox_media.json", 32, 32, 4), ("classroom_fog", "scenes/classroom/vision_scene.json", 48, 27, 2),
         ("cbox_lights", "scenes/cbox/cbox_lights.json", 32, 32, 4), ("cbox_sinc", "scenes/cbox/cbox_sinc.json", 32, 32, 4),
         ("glass_of_water", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),
         ("cbox_power", "scenes/cbox/cbox_power.json", 32, 32, 4), ("cbox_sheen", "scenes/cbox/cbox_sheen.json", 32, 32, 4), ("cbox_extra", "scenes/cbox/cbox_extra.json", 32, 32, 4),
         # Material::compute_shading_frame with "normal" slots, mix / add with a principled_bsdf child (LobeSet::flatten), shape/sphere
         ("cbox_normal", "scenes/cbox/cbox_normal.json", 32, 32, 4),
         # the reference's playground scene as shipped: a "multiply" shader node (checker.jpg x a constant), a normal map, mix, principled
         ("playground", "scenes/playground/vision_scene.json", 48, 48, 2),
         # a complete shipped scene (no stripped asset): 19 meshes, 13 lights, 3 JPG textures, 21 materials
         ("staircase2", "scenes/staircase2/vision_scene.json", 48, 27, 2),
         # the reference's default Cornell scene as shipped: spot + point + spherical + projector (Painting3.jpg) + area light in one scene
         ("cbox_vision_scene", "scenes/cbox/vision_scene.json", 48, 48, 2),
         # spectrum/hero: every material family incl. dispersive BK7 glass + measured Cu, textures; diffuse only;
         # media; point + spot lights; config 4 "spectral glass"; classroom with its environment map and textures
         ("cbox_hero", "scenes/cbox/cbox_hero.json", 32, 32, 4), ("cbox_hero_matte", "scenes/cbox/cbox_hero_matte.json", 32, 32, 4),
         ("cbox_hero_media", "scenes/cbox/cbox_hero_media.json", 32, 32, 4), ("cbox_hero_lights", "scenes/cbox/cbox_hero_lights.json", 32, 32, 4),
         ("glass_of_water_hero", "scenes/glass-of-water/vision_scene.json", 48, 48, 2), ("classroom_hero", "scenes/classroom/vision_scene.json", 48, 27, 2),
         # spectrum/hero with "dimension": 4 (cbox-prism.json:692-697): four wavelengths per path — the vmk_hero4.hip instance / the ORC_SPEC_DIM = 4 oracle build
         ("cbox_hero4", "scenes/cbox/cbox_hero.json", 32, 32, 4), ("cbox_hero4_matte", "scenes/cbox/cbox_hero_matte.json", 32, 32, 4),
         ("cbox_hero4_media", "scenes/cbox/cbox_hero_media.json", 32, 32, 4), ("glass_of_water_hero4", "scenes/glass-of-water/vision_scene.json", 48, 48, 2),
         ("cbox_prism_hero4", "scenes/cbox/cbox-prism.json", 48, 48, 2)]  # the reference's own dimension-4 scene, as shipped
MEDIA = {"cbox_media", "classroom_fog", "cbox_hero_media", "cbox_hero4_media"}  # rendered with the scene's "mediums" block honoured
SPECTRUM = {"glass_of_water_hero": "hero", "classroom_hero": "hero",  # shipped scenes with the spectrum forced to hero (vmk_host_options.spectrum)
            "cbox_hero4": "hero4", "cbox_hero4_matte": "hero4", "cbox_hero4_media": "hero4", "glass_of_water_hero4": "hero4"}  # ... to hero with four wavelengths
ONLY = set(sys.argv[1:])  # optional: regenerate the named cases only
for name, path, w, h, spp in CASES:
    if ONLY and name not in ONLY: continue
    hs = HostScene(os.path.join(ROOT, path), width=w, height=h, mediums=name in MEDIA, spectrum=SPECTRUM.get(name))
    img, cnt = oracle_py.OracleScene(hs).render(hs.params_copy(), 0, spp)
    out = os.path.join(ROOT, "tests", "golden", f"{name}_{w}x{h}x{spp}.npy")
    np.save(out, img)
    print(out, img[..., :3].mean(), cnt)

#!/usr/bin/env python3
"""vmk_self_check (megakernel variant vs unit kernel) on a few scenes, for the library VMK_LIB points at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline
for scene, w, h, kw in (("scenes/cbox/cbox_matte.json", 32, 32, {}), ("scenes/cbox/cbox_materials.json", 32, 32, {}),
                        ("scenes/cbox/cbox_media.json", 32, 32, {"mediums": True}), ("scenes/classroom/vision_scene.json", 64, 36, {}),
                        ("scenes/classroom/vision_scene.json", 64, 36, {"mediums": True}), ("scenes/cbox/cbox_lights.json", 32, 32, {}),
                        ("scenes/cbox/cbox_extra.json", 32, 32, {}), ("scenes/glass-of-water/vision_scene.json", 48, 48, {})):
    pipe = Pipeline(os.path.join(ROOT, scene), width=w, height=h, **kw)
    pipe.prepare()
    try:
        n = pipe.backend.self_check()
        print(os.environ.get("VMK_LIB", "default"), scene, "ok", n)
    except Exception as e:
        print(os.environ.get("VMK_LIB", "default"), scene, "MISMATCH", str(e)[:120])
    pipe.close()

import os, sys
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline
for scene, w, h, kw in (("scenes/cbox/cbox_matte.json", 32, 32, {}), ("scenes/cbox/cbox_materials.json", 32, 32, {}),
                        ("scenes/cbox/cbox_media.json", 32, 32, {"mediums": True}), ("scenes/cbox/cbox_lights.json", 32, 32, {}),
                        ("scenes/cbox/cbox_extra.json", 32, 32, {}), ("scenes/classroom/vision_scene.json", 64, 36, {}),
                        ("scenes/classroom/vision_scene.json", 64, 36, {"mediums": True}), ("scenes/glass-of-water/vision_scene.json", 48, 48, {})):
    pipe = Pipeline(os.path.join(ROOT, scene), width=w, height=h, **kw)
    try:
        pipe.prepare(self_check=False)  # the explicit call below is the experiment
        pipe.backend.set_auto_self_check(False)
        n = pipe.backend.self_check()
        print(scene, kw, "ok", n, flush=True)
    except Exception as e:
        print(scene, kw, "MISMATCH", str(e)[:200], flush=True)
        sys.exit(3)  # stop at the first mismatch: a miscompiled variant is not run on the larger scenes
    pipe.close()

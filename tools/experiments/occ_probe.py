import os, sys
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline
for scene, w, h, spp in (("scenes/classroom/vision_scene.json", 1920, 1080, 32), ("scenes/cbox/cbox_matte.json", 1024, 1024, 32)):
    pipe = Pipeline(os.path.join(ROOT, scene), width=w, height=h)
    pipe.prepare(self_check=False); pipe.backend.set_auto_self_check(False)
    pipe.invalidate(); pipe.backend.reset_counters()
    ms = pipe.render(frames=spp)
    print(scene, ms, pipe.counters(), flush=True)
    pipe.close()

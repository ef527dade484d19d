"""Wave-time profile of the megakernel from a probe build (clock64() at the convergent points of the vertex loop; counters hijacked:
closest_rays = closest traversal, surface_hits = miss / emitter / interaction / NEE sampling up to the shadow ray, shadow_rays = shadow
traversal, paths = path regeneration, nodes_visited / 64 = whole path_bounce, tris_tested / 64 = film write; units of 64 cycles summed over
waves).  usage (GPU box): VMK_LIB=vision_amd/lib/exp/libvmk_tprobe.so python tools/experiments/time_probe.py"""
import os, sys
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from vision_amd.pipeline import Pipeline
for scene, w, h, spp in (("scenes/classroom/vision_scene.json", 1920, 1080, 32), ("scenes/cbox/cbox_matte.json", 1024, 1024, 32), ("scenes/bathroom2/vision_scene.json", 1920, 1080, 16)):
    if not os.path.exists(os.path.join(ROOT, scene)):
        continue
    pipe = Pipeline(os.path.join(ROOT, scene), width=w, height=h)
    pipe.prepare(self_check=False); pipe.backend.set_auto_self_check(False)
    pipe.backend.set_traversal_counters(False)
    pipe.invalidate(); pipe.backend.reset_counters()
    ms = pipe.render(frames=spp)
    c = pipe.counters()
    bounce = c["nodes_visited"] / 64.0
    seg = {"regen": c["paths"], "closest traversal": c["closest_rays"], "hit -> shadow ray (miss, emitter, interaction, NEE sample)": c["surface_hits"],
           "shadow traversal": c["shadow_rays"], "material + evaluate/sample + Ld + RR + spawn": bounce - c["closest_rays"] - c["surface_hits"] - c["shadow_rays"],
           "film write / terminate": c["tris_tested"] / 64.0}
    tot = sum(seg.values())
    print(f"{scene} {w}x{h} {spp} spp: kernel {ms:.1f} ms; wave-time {tot * 64 / 1e9:.2f} Gcycles")
    for k, v in seg.items():
        print(f"   {100 * v / tot:5.1f} %  {k}")
    pipe.close()

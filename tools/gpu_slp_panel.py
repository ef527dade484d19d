#!/usr/bin/env python3
"""vmk_self_check on cbox_matte (the scene that shows the SLP-vectoriser miscompile, profiles/r03_slp_root_cause.md) for every
vision_amd/lib/exp/libvmk_slp*.so, each in its own child process (VMK_LIB is read at import).  usage (GPU box): python tools/gpu_slp_panel.py"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import os, sys
sys.path.insert(0, %r)
from vision_amd.pipeline import Pipeline
pipe = Pipeline(os.path.join(%r, "scenes/cbox/cbox_matte.json"), width=32, height=32)
pipe.prepare(self_check=False)
try:
    print("ok", pipe.backend.self_check())
except Exception as e:
    print("MISMATCH", str(e)[:90])
""" % (ROOT, ROOT)
for lib in sorted(glob.glob(os.path.join(ROOT, "vision_amd", "lib", "exp", "libvmk_slp*.so"))):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, VMK_LIB=lib), capture_output=True, text=True, timeout=120)
    print(os.path.basename(lib), (r.stdout.strip() or r.stderr.strip()[-200:]), flush=True)

"""vision_amd — MI355X-native megakernel path-tracing backend behind Vision's Integrator/Pipeline plugin surface.

Layout: csrc/device (hand-written HIP for gfx950 + the C-ABI of include/vmk.h), csrc/host (C++ scene front-end,
include/vmk_host.h), and this thin Python mirror of the reference's Pipeline/Integrator interface.
"""
from .pipeline import Pipeline, PathTracingIntegrator, FrameBuffer  # noqa: F401
from .backend import Backend, BackendError  # noqa: F401
from .host import HostScene, HostError  # noqa: F401

"""Host-side mirror of Vision's Pipeline / Integrator / FrameBuffer surface for the path-tracing hot path.

Names, argument meaning and call order follow the reference (Vision `src/`):
  Pipeline            base/mgr/pipeline.h:80-103, pipelines/fixed/pipeline.cpp:14-42   (init_scene, prepare, compile,
                      render, display, invalidate, final_picture, save_result, change_resolution)
  PathTracingIntegrator  render_core/integrator/pt.cpp:27-37,53-81,96-116; base/integral/integrator.h:33-77,122-196
                      (prepare, compile, render, invalidation, frame_index, max_depth/min_depth/rr_threshold/mis_mode)
  FrameBuffer         base/sensor/frame_buffer.cpp:15-38,117-154   (resolution, exposure, accumulation buffer, tone mapping)
What differs by design: `render(frames=N)` renders a whole batch of N 1-spp frames in one megakernel launch (the
reference issues 4 kernels + a host sync per frame), and the film lives in HBM until downloaded.
All compute goes through the C-ABI (include/vmk.h); there is no CPU fallback.
"""
import os
import time

import numpy as np

from . import _abi
from .backend import Backend, BackendError
from .host import HostScene


class FrameBuffer:
    """frame_buffer.cpp:15-26 — resolution / exposure / tone mapper come from pipeline.param.frame_buffer.param."""

    def __init__(self, pipeline):
        self._p = pipeline

    @property
    def resolution(self):
        return (self._p.params.width, self._p.params.height)

    @property
    def exposure(self):
        return self._p.params.exposure

    def pixel_num(self):
        return self._p.params.width * self._p.params.height

    def download(self):
        """Linear accumulation buffer (accumulation_buffer of the reference), float32 [H, W, 4]."""
        if self._p._torch_fb is not None:
            self._p.backend.synchronize()
            return self._p._torch_fb.detach().cpu().numpy()
        return self._p.backend.download_accum()

    def tone_mapped(self, final_picture=False):
        """exposure -> tone map (K4 of pt.cpp:96-116); final_picture=True adds the second tone map + sRGB of
        Pipeline::final_picture (pipeline.cpp:337-354, postprocessor.cpp:13-30)."""
        return self._p.backend.tonemap(final_picture)


class PathTracingIntegrator:
    """integrator/pt — keeps the reference's parameter names and frame counter semantics."""

    def __init__(self, pipeline):
        self._p = pipeline
        self._frame_index = 0
        self._render_time = 0.0

    max_depth = property(lambda self: self._p.params.max_depth)
    min_depth = property(lambda self: self._p.params.min_depth)
    rr_threshold = property(lambda self: self._p.params.rr_threshold)
    mis_mode = property(lambda self: self._p.params.mis_mode)

    def frame_index(self):
        return self._frame_index

    def render_time(self):
        return self._render_time

    def prepare(self):
        pass  # parameters travel inside vmk_render_params (EncodedData upload in the reference)

    def compile(self):
        pass  # kernels are compiled ahead of time for gfx950 (no JIT)

    def invalidation(self):  # integrator.cpp:13-18
        self._frame_index = 0
        self._render_time = 0.0
        self._p.backend.reset_accum()

    def render(self, frames=1, tiles=None, timed=True):
        """Render `frames` consecutive 1-spp frames [frame_index, frame_index+frames) fused with accumulation."""
        ms = self._p.backend.render_batch(self._frame_index, frames, tiles=tiles, timed=timed)
        self._frame_index += frames
        if ms is not None:
            self._render_time += ms
        return ms


class Pipeline:
    """pipeline/fixed.  `Pipeline(scene_json)` plays Importer::import_scene + init_scene."""

    def __init__(self, scene_file, device=0, width=0, height=0, max_depth=-1, min_depth=-1, procedural_env=True,
                 drop_unsupported_lights=False, mediums=False, spectrum=None, missing_assets=None):
        # every option of the scene load is kept: change_resolution re-derives the tables with the same ones
        self._host_options = dict(max_depth=max_depth, min_depth=min_depth, procedural_env=procedural_env,
                                  drop_unsupported_lights=drop_unsupported_lights, mediums=mediums, spectrum=spectrum,
                                  missing_assets=missing_assets)
        self.host_scene = HostScene(scene_file, width=width, height=height, **self._host_options)
        self.params = self.host_scene.params_copy()
        self.backend = Backend(device)
        self.frame_buffer = FrameBuffer(self)
        self.integrator = PathTracingIntegrator(self)
        self.accel_info = None
        self.tiles = None
        self._torch_fb = None
        self._prepared = False

    # ---- FixedRenderPipeline::prepare (fixed/pipeline.cpp:14-23): upload + build accel ----
    def prepare(self, self_check=None):
        self.backend.upload_scene(self.host_scene)
        self.accel_info = self.backend.build_accel()
        self.backend.set_render_params(self.params)
        self.integrator.prepare()
        self._prepared = True
        # optional toolchain self-check (megakernel variant vs unit kernel, include/vmk.h): on by argument or VMK_SELF_CHECK=1
        if self_check if self_check is not None else os.environ.get("VMK_SELF_CHECK") == "1":
            self.backend.self_check()

    def compile(self):
        self.integrator.compile()

    def render_aov(self, frame=0):
        """G-buffer planes of the primary hit (FrameBuffer::compute_GBuffer, frame_buffer.cpp:156-219,322-339): normal,
        albedo, emission [H, W, 4] and linear depth [H, W] — the denoiser hand-off of the reference."""
        return self.backend.render_aov(frame)

    def change_resolution(self, width, height):
        """Pipeline::change_resolution: re-derives the camera matrices for the new film size (sensor.cpp:58-71)."""
        path = self.host_scene.json_path
        self.host_scene.close()
        self.host_scene = HostScene(path, width=width, height=height, **self._host_options)
        self.params = self.host_scene.params_copy()
        # vmk_set_render_params falls back to a ctx-owned film of the new size: a caller-owned tensor of the old size is
        # no longer bound (use_torch_framebuffer again with a tensor of the new shape)
        self._torch_fb = None
        self.backend.set_render_params(self.params)
        self.invalidate()

    def set_tiles(self, tile_size, rank, world):
        """Multi-GPU sharding: this process renders only the tiles it owns (include/vmk.h vmk_tiles)."""
        self.tiles = _abi.Tiles(tile_size, rank, world) if world > 1 else None

    def use_torch_framebuffer(self, tensor):
        """Render into a caller-owned CUDA tensor [H, W, 4] float32 (so torch.distributed can all-reduce it)."""
        assert tensor.is_cuda and tensor.is_contiguous() and tuple(tensor.shape) == (self.params.height, self.params.width, 4)
        self._torch_fb = tensor
        self.backend.set_framebuffer(tensor.data_ptr())

    def invalidate(self):
        self.integrator.invalidation()

    def render(self, dt=0.0, frames=1, timed=True):
        if not self._prepared:
            raise BackendError("Pipeline.render before prepare()")
        return self.integrator.render(frames=frames, tiles=self.tiles, timed=timed)

    def display(self, dt=0.0):
        return self.render(dt)

    def frame_index(self):
        return self.integrator.frame_index()

    def counters(self):
        return self.backend.counters()

    def final_picture(self, fn=None):
        """Pipeline::final_picture (pipeline.cpp:337-354): the tone-mapped output buffer through the tone mapper a second time, and
        through the sRGB curve unless the output name ends in exr / hdr."""
        from .host import final_picture_mode
        return self.backend.tonemap(final_picture_mode(fn) if fn else 1)

    def save_result(self, fn=None):
        """Pipeline::save_result (pipeline.cpp:190-204): final_picture -> Image::save_image, both in the C++ host / on the device
        (vmk_tonemap + vmk_host_save_image: 8-bit PNG, float32 OpenEXR, Radiance HDR).  A ".npy" name dumps the linear buffer."""
        from .host import save_image
        fn = fn or self.host_scene.output_fn
        if fn.endswith(".npy"):
            np.save(fn, self.frame_buffer.download())
            return fn
        return save_image(fn, self.final_picture(fn))

    def close(self):
        if getattr(self, "backend", None):
            self.backend.close()
        if getattr(self, "host_scene", None):
            self.host_scene.close()

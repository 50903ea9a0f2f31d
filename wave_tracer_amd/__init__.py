"""wave_tracer_amd — MI355X-native implementation of wave_tracer's per-sample wave-optical integrator path.

The compute path is the HIP library ``libwtgpu.so`` (hand-written gfx950 kernels behind the C-ABI of
``include/wtgpu.h``); this package is the thin host-side mirror of the reference's render-loop interface
(``scene_renderer_t``: src/scene/render.cpp:381-579) on top of it.  PyTorch is used for device memory, streams and
``torch.distributed`` (RCCL) only.
"""
from .api import Scene, SceneParams, WtgpuError, lib_path, load_library  # noqa: F401
from .render import render, develop, render_distributed  # noqa: F401

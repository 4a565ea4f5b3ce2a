"""MI355X-native rendering core for the reference ray tracer's per-pixel ray/scene loop.

  raytracer_amd.capi      ctypes binding of the C ABI (include/rt_mi355.h, csrc/librt_mi355.so)
  raytracer_amd.host_api  ctypes binding of the reference's own host API (host/libraytracer_host.so)
  raytracer_amd.sharding  frame sharding over ranks and the framebuffer gather (torch.distributed)

The compute path is the HIP library; Python only moves pointers.
"""
from . import capi, host_api  # noqa: F401

__all__ = ["capi", "host_api"]

"""MI355X-native (gfx950) 3DGS training rasterizer behind the reference's gsplat `Ops.h` API.

Layout:
  csrc/       hand-written HIP kernels + the C ABI (include/lfs_gsplat.h) -> liblfs_gsplat.so
  capi.py     ctypes binding of the C ABI (raw device pointers of torch tensors)
  ops.py      Python mirror of `namespace gsplat` (gsplat/Ops.h) + fast_gs::optimizer::adam_step_wrapper
  rasterizer.py  mirror of src/training/rasterization/{rasterizer,rasterizer_autograd}.cpp
  fused_adam.py  mirror of src/training/optimizers/fused_adam.cpp
  scenes.py   synthetic scenes of SURVEY.md §8d (SYN-A .. SYN-D)
  dist.py     data-parallel view sharding + RCCL gradient all-reduce

There is NO CPU fallback: every op raises if the HIP library is missing or a tensor is not on
the GPU.  The CPU oracle lives in /oracle and is never imported from here.
"""
from . import capi  # noqa: F401
from .capi import CameraModelType, ShutterType, UnscentedTransformParameters, library_path, load_library  # noqa: F401

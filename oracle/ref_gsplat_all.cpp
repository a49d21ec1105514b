// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// The reference's WHOLE gsplat library - every .cu (kernels and their launch functions) and every .cpp (the operators of gsplat/Ops.h: argument checks, output
// allocation, the two-pass intersection with cumsum and radix sort, the channel-count dispatch of the rasterizer) - compiled in place as ONE host translation unit
// and run on the CPU (oracle/Makefile, `make refgsplat`). The .cu files pass through oracle/ref_cu_prep.py (launch syntax, the saturating float -> unsigned
// conversion) and Common.h through sed (CHECK_CUDA's `x.is_cuda()` -> true: the tensors live on the CPU here) into a scratch directory; nothing else is edited,
// nothing is copied into the repository. CUDA's execution model: oracle/ref_emul/. Linked with ref_raster_shim.cpp (-DREF_REAL_GSPLAT) and the reference's
// rasterizer.cpp / rasterizer_autograd.cpp / camera.cpp into oracle/_ref/libref_raster_full.so: the render + backward of a training step with NOTHING of the
// operator layer restated. tests/test_oracle_ref_raster_golden.py checks that it reproduces tests/golden/ref_raster.npz (generated through the restated launch
// sequences of ref_raster_shim.cpp / ref_kernels.cpp), which validates those restatements as well.
#include "cuda_runtime.h" // ref_emul: the emulator + cuemu::launcher
#undef __shared__          // (cuda_runtime.h makes it `static` for the fastgs kernels' static arrays; the gsplat kernels only declare `extern __shared__ int s[]`)
#define __shared__
#include <ATen/Dispatch.h>
#include <ATen/Functions.h>
#include <ATen/core/Tensor.h>
#include <torch/torch.h>

#include "ProjectionUT3DGSFused.cu"
#include "SphericalHarmonicsCUDA.cu"
#include "IntersectTile.cu"
#include "RasterizeToPixelsFromWorld3DGSFwd.cu"
#include "RasterizeToPixelsFromWorld3DGSBwd.cu"
#include "RelocationCUDA.cu"
#include "QuatToRotmatCUDA.cu"

namespace gsplat {
alignas(64) int s[1 << 18]; // `extern __shared__ int s[]` of the rasterizer kernels: 1 MiB, one workgroup runs at a time
}

#include "Projection.cpp"
#include "SphericalHarmonics.cpp"
#include "Intersect.cpp"
#include "Rasterization.cpp"
#include "Relocation.cpp"
#include "QuatToRotmat.cpp"

// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. One translation unit under the emulator for `make reffast`: the reference's fastgs forward.cu / backward.cu (host
// sequences + kernels, through the sed edits of `make refk_fastgs`) and its libtorch wrapper rasterization_api.cu (sed: torch::kCUDA -> torch::kCPU), in place.
#include <torch/torch.h>  // before __CUDACC__ is defined: ATen would take it for a real CUDA compiler and ask for <cuda.h>
#include <functional>
#include <stdexcept>
#include <tuple>
#define __CUDACC__ 1   // helper_math.h: skip its host re-definitions of fminf / fmaxf / min / max / rsqrtf (math.h and cuda_emul.h provide them)
#include "cuda_runtime.h"
#include "cooperative_groups.h"
#include "k_fastgs_forward.inc"
#include "k_fastgs_backward.inc"
#include "rasterization_api.cu"

// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. One translation unit under the emulator for `make reflosshost`: the reference's loss-side .cu files WHOLE - kernels and the
// host functions below them (fusedssim / fusedssim_backward of ssim.cu; slice_forward_cuda, slice_backward_cuda, tv_loss_forward_cuda, tv_loss_backward_cuda of
// bilateral_grid_{forward,backward,tv}.cu) - compiled in place through oracle/ref_cu_prep.py (the launch syntax only).
#include <torch/torch.h>
#include "cuda_runtime.h" // ref_emul: the emulator, `__shared__` = static for the kernels' static tiles, cuemu::launcher
#include "cooperative_groups.h"
#include "ssim.cu"
#include "bilateral_grid_forward.cu"
#include "bilateral_grid_backward.cu"
#include "bilateral_grid_tv.cu"

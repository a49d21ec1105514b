// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// The reference's OWN loss-side kernels run on the CPU: src/training/kernels/ssim.cu:10-424 (fusedssimCUDA, fusedssim_backwardCUDA),
// bilateral_grid_forward.cu:8-94, bilateral_grid_backward.cu:8-153, bilateral_grid_tv.cu:9-142 - the kernel parts, compiled IN PLACE as host C++ by
// `make -C oracle refk_loss` (the recipe pipes the line ranges into a scratch directory that is deleted after the compile; the libtorch wrappers below them use
// torch types and the <<<...>>> syntax; nothing of the reference is copied into the repository, oracle/_ref only receives the .so). The launch configurations
// of those wrappers are restated here, each with its citation. CUDA's execution model: oracle/ref_emul/ (fibers, static __shared__ arrays, cub::BlockReduce).
// oracle/make_golden_refk_loss.py writes tests/golden/refk_loss.npz, against which the torch-side restatements used as oracle for these rows
// (tests/ssim_reference.py, tests/test_oracle_bilateral.py) and the HIP kernels (csrc/ssim.hip, bilateral_grid.hip) are both checked.
#define __CUDACC__ 1
#include "cuda_runtime.h"
#include "cooperative_groups.h"
#include "cub/cub.cuh"
#include "k_ssim.inc"
#include "k_bilateral_fwd.inc"
#include "k_bilateral_bwd.inc"
#include "k_bilateral_tv.inc"

#include <algorithm>

#define REFK_API extern "C" __attribute__((visibility("default")))

// fusedssim (ssim.cu:430-468): grid = (ceil(W/16), ceil(H/16), B), block = (16, 16); outputs zero-initialised
REFK_API void refk_fusedssim(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, int train, float* ssim_map, float* dm_dmu1,
                             float* dm_dsigma1_sq, float* dm_dsigma12) {
    const size_t n = size_t(B) * CH * H * W;
    std::fill(ssim_map, ssim_map + n, 0.f);
    if (train) { std::fill(dm_dmu1, dm_dmu1 + n, 0.f); std::fill(dm_dsigma1_sq, dm_dsigma1_sq + n, 0.f); std::fill(dm_dsigma12, dm_dsigma12 + n, 0.f); }
    const dim3 grid((W + BLOCK_X - 1) / BLOCK_X, (H + BLOCK_Y - 1) / BLOCK_Y, B), block(BLOCK_X, BLOCK_Y);
    cuemu::launcher(fusedssimCUDA, grid, block)(H, W, CH, C1, C2, img1, img2, ssim_map, train ? dm_dmu1 : nullptr, train ? dm_dsigma1_sq : nullptr,
                                                train ? dm_dsigma12 : nullptr);
}
// fusedssim_backward (ssim.cu:476-510)
REFK_API void refk_fusedssim_backward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, const float* dL_dmap,
                                      const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1) {
    std::fill(dL_dimg1, dL_dimg1 + size_t(B) * CH * H * W, 0.f);
    const dim3 grid((W + BLOCK_X - 1) / BLOCK_X, (H + BLOCK_Y - 1) / BLOCK_Y, B), block(BLOCK_X, BLOCK_Y);
    cuemu::launcher(fusedssim_backwardCUDA, grid, block)(H, W, CH, C1, C2, img1, img2, dL_dmap, dL_dimg1, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
}
// slice_forward_cuda (bilateral_grid_forward.cu:96-115): 256 threads, ceil(h w / 256) blocks
REFK_API void refk_bilateral_slice_forward(const float* grid, const float* rgb, float* output, int L, int H, int W, int h, int w) {
    cuemu::launcher(gs::bilateral_grid::slice_forward_kernel, (h * w + 255) / 256, 256)(grid, rgb, output, L, H, W, h, w);
}
// slice_backward_cuda (bilateral_grid_backward.cu:155-184): grad_grid zero-initialised, min(ceil(h w / 256), 65535) blocks
REFK_API void refk_bilateral_slice_backward(const float* grid, const float* rgb, const float* grad_output, float* grad_grid, float* grad_rgb, int L, int H, int W,
                                            int h, int w) {
    std::fill(grad_grid, grad_grid + size_t(12) * L * H * W, 0.f);
    cuemu::launcher(gs::bilateral_grid::slice_backward_kernel, std::min((h * w + 255) / 256, 65535), 256)(grid, rgb, grad_output, grad_grid, grad_rgb, L, H, W, h, w);
}
// tv_loss_forward_cuda / tv_loss_backward_cuda (bilateral_grid_tv.cu:144-188): min(ceil(total / 256), 2048) blocks of 256
REFK_API void refk_bilateral_tv_forward(const float* grids, float* tv_loss, int N, int L, int H, int W) {
    *tv_loss = 0.f;
    const int total = N * L * H * W;
    cuemu::launcher(gs::bilateral_grid::tv_loss_forward_kernel, std::min((total + 255) / 256, 2048), 256)(grids, tv_loss, N, L, H, W);
}
REFK_API void refk_bilateral_tv_backward(const float* grids, float grad_output, float* grad_grids, int N, int L, int H, int W) {
    const size_t total = size_t(N) * 12 * L * H * W;
    std::fill(grad_grids, grad_grids + total, 0.f);
    cuemu::launcher(gs::bilateral_grid::tv_loss_backward_kernel, std::min(int((total + 255) / 256), 2048), 256)(grids, grad_output, grad_grids, N, L, H, W);
}

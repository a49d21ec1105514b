// Fused SSIM (11-tap Gaussian window, sigma 1.5, zero padding) forward / backward and the trainer's photometric loss
// (1 - lambda) * L1 + lambda * (1 - mean SSIM) — SURVEY.md §8f row 2. Replaces fusedssim / fusedssim_backward
// (reference: src/training/kernels/ssim.cu:64-282 forward, :284-426 backward, host :430-510; autograd wrapper
// include/kernels/fused_ssim.cuh:30-107 incl. the "valid" crop of 5 pixels; loss src/training/trainer.cpp:122-125).
//
// HBM/LDS-bound 2-D stencil. One 16x16-pixel workgroup (4 wavefronts) stages a 26x26 halo tile in LDS, runs the
// separable window as a horizontal pass over 26 rows (5 running moments per pixel) and a vertical pass over the 16
// output rows, channel by channel. Two things differ from the reference's structure:
//   * img1 may be handed over in the rasterizer's HWC layout, un-clamped: the clamp to [0,1] (rasterizer.cpp:399), the
//     CHW view, the L1 term, the crop mask, the loss reduction and the clamp's gradient mask are folded into the two
//     kernels, so the loss costs two passes over the image instead of the ~12 libtorch passes of trainer.cpp:118-126;
//   * in that mode the forward stores the three partial-derivative maps already multiplied by dL/dmap (a constant inside
//     the crop, 0 outside), so the backward reads 3 planes per channel instead of 4.
#include "lfs_math.cuh"
#include "lfs_prof.h"
#include "../../include/lfs_gsplat.h"

namespace lfs {

constexpr int SS_B = 16, SS_HALO = 5, SS_T = SS_B + 2 * SS_HALO; // 26

// exp(-(i-5)^2 / (2 * 1.5^2)) / sum, as float32 (the window every fused-ssim implementation ships)
__constant__ float c_gauss[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                                  0.21300552785396576f,  0.26601171493530273f,   0.21300552785396576f,  0.10936068743467331f,
                                  0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f};

struct SsimImg {
    const float* p; int hwc_clamp; // 0: [B,CH,H,W] as is; 1: a rasterizer output handed to the fused loss:
    int chw;                       //    layout [B,H,W,CH] (3DGUT rasterizer) or, with chw != 0, [B,CH,H,W] (fastgs rasterizer),
    int clamp;                     //    clamped to [0,1] on the way in (and the gradient masked) when clamp != 0
};
LFS_DI size_t ssim_index(const SsimImg& im, int b, int c, int y, int x, int CH, int H, int W) {
    return (im.hwc_clamp && !im.chw) ? ((size_t(b) * H + y) * W + x) * CH + c : ((size_t(b) * CH + c) * H + y) * W + x;
}
LFS_DI float ssim_fetch(const SsimImg& im, int b, int c, int y, int x, int CH, int H, int W) {
    if (x < 0 || x >= W || y < 0 || y >= H) return 0.f; // zero padding
    const float v = im.p[ssim_index(im, b, c, y, x, CH, H, W)];
    return (im.hwc_clamp && im.clamp) ? fminf(fmaxf(v, 0.f), 1.f) : v;
}

struct SsimFwdArgs {
    int H, W, CH; float C1, C2;
    SsimImg img1; const float* img2;            // img2 always [B,CH,H,W]
    float* ssim_map;                            // [B,CH,H,W] or NULL
    float* dm_dmu1; float* dm_dsigma1_sq; float* dm_dsigma12; // [B,CH,H,W] or NULL (inference)
    // fused-loss mode (fused != 0): derivative maps are stored pre-multiplied by dL/dmap = in_crop ? chain : 0 and
    // loss[0] += w_ssim * sum_crop(ssim) + w_l1 * sum |img1 - img2| + loss_const
    int fused; float chain; int crop; float* loss; float w_ssim, w_l1, loss_const;
};

__global__ void __launch_bounds__(256) ssim_fwd_kernel(const SsimFwdArgs a) {
    __shared__ float s_tile[SS_T][SS_T][2];
    __shared__ float s_conv[SS_T][SS_B][5]; // after the horizontal pass: E[x], E[x^2], E[y], E[y^2], E[xy]
    __shared__ float s_red[2][4];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * SS_B, y0 = blockIdx.y * SS_B;
    const int px = x0 + tx, py = y0 + ty;
    const bool in_img = px < a.W && py < a.H;
    const bool in_crop = !a.crop || (px >= SS_HALO && px < a.W - SS_HALO && py >= SS_HALO && py < a.H - SS_HALO);
    float acc_ssim = 0.f, acc_l1 = 0.f;
    for (int c = 0; c < a.CH; ++c) {
        for (int i = threadIdx.x; i < SS_T * SS_T; i += 256) {
            const int ly = i / SS_T, lx = i % SS_T;
            const int gy = y0 + ly - SS_HALO, gx = x0 + lx - SS_HALO;
            s_tile[ly][lx][0] = ssim_fetch(a.img1, b, c, gy, gx, a.CH, a.H, a.W);
            s_tile[ly][lx][1] = (gx < 0 || gx >= a.W || gy < 0 || gy >= a.H) ? 0.f : a.img2[((size_t(b) * a.CH + c) * a.H + gy) * a.W + gx];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SS_T * SS_B; i += 256) { // horizontal pass: 26 rows x 16 columns
            const int ly = i / SS_B, lx = (i % SS_B) + SS_HALO;
            float sx = 0.f, sxx = 0.f, sy = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int d = 1; d <= SS_HALO; ++d) { // symmetric pairs first, centre last (the reference's summation order)
                const float w = c_gauss[SS_HALO - d];
                const float xl = s_tile[ly][lx - d][0], yl = s_tile[ly][lx - d][1], xr = s_tile[ly][lx + d][0], yr = s_tile[ly][lx + d][1];
                sx += (xl + xr) * w; sxx += (xl * xl + xr * xr) * w;
                sy += (yl + yr) * w; syy += (yl * yl + yr * yr) * w;
                sxy += (xl * yl + xr * yr) * w;
            }
            const float wc = c_gauss[SS_HALO], xc = s_tile[ly][lx][0], yc = s_tile[ly][lx][1];
            sx += xc * wc; sxx += xc * xc * wc; sy += yc * wc; syy += yc * yc * wc; sxy += xc * yc * wc;
            float* o = s_conv[ly][lx - SS_HALO];
            o[0] = sx; o[1] = sxx; o[2] = sy; o[3] = syy; o[4] = sxy;
        }
        __syncthreads();
        {   // vertical pass + SSIM for this thread's pixel
            const int ly = ty + SS_HALO;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, o4 = 0.f;
#pragma unroll
            for (int d = 1; d <= SS_HALO; ++d) {
                const float w = c_gauss[SS_HALO - d];
                const float* t = s_conv[ly - d][tx]; const float* u = s_conv[ly + d][tx];
                o0 += (t[0] + u[0]) * w; o1 += (t[1] + u[1]) * w; o2 += (t[2] + u[2]) * w; o3 += (t[3] + u[3]) * w; o4 += (t[4] + u[4]) * w;
            }
            const float wc = c_gauss[SS_HALO]; const float* m = s_conv[ly][tx];
            o0 += m[0] * wc; o1 += m[1] * wc; o2 += m[2] * wc; o3 += m[3] * wc; o4 += m[4] * wc;
            if (in_img) {
                const float mu1 = o0, mu2 = o2;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
                const float sigma1_sq = o1 - mu1_sq, sigma2_sq = o3 - mu2_sq, sigma12 = o4 - mu1 * mu2;
                const float A = mu1_sq + mu2_sq + a.C1, B = sigma1_sq + sigma2_sq + a.C2;
                const float Cn = 2.f * mu1 * mu2 + a.C1, Dn = 2.f * sigma12 + a.C2;
                const float val = (Cn * Dn) / (A * B);
                const size_t gi = ((size_t(b) * a.CH + c) * a.H + py) * a.W + px;
                if (a.ssim_map) a.ssim_map[gi] = val;
                if (a.dm_dmu1) {
                    float d_mu1 = ((mu2 * 2.f * Dn) / (A * B) - (mu2 * 2.f * Cn) / (A * B) - (mu1 * 2.f * Cn * Dn) / (A * A * B) + (mu1 * 2.f * Cn * Dn) / (A * B * B));
                    float d_s1 = (-Cn * Dn) / (A * B * B);
                    float d_s12 = (2.f * Cn) / (A * B);
                    if (a.fused) { const float ch = in_crop ? a.chain : 0.f; d_mu1 *= ch; d_s1 *= ch; d_s12 *= ch; }
                    a.dm_dmu1[gi] = d_mu1; a.dm_dsigma1_sq[gi] = d_s1; a.dm_dsigma12[gi] = d_s12;
                }
                if (a.loss) {
                    if (in_crop) acc_ssim += val;
                    acc_l1 += fabsf(s_tile[ty + SS_HALO][tx + SS_HALO][0] - s_tile[ty + SS_HALO][tx + SS_HALO][1]);
                }
            }
        }
        __syncthreads();
    }
    if (a.loss) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { acc_ssim += __shfl_xor(acc_ssim, m, 64); acc_l1 += __shfl_xor(acc_l1, m, 64); }
        if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = acc_ssim; s_red[1][threadIdx.x >> 6] = acc_l1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = a.w_ssim * (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]) + a.w_l1 * (s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]);
            if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) v += a.loss_const;
            atomicAdd(a.loss, v);
        }
    }
}

struct SsimBwdArgs {
    int H, W, CH;
    SsimImg img1; const float* img2;
    const float* dL_dmap;                       // [B,CH,H,W], or NULL when the maps below are pre-multiplied (fused-loss mode)
    const float* dm_dmu1; const float* dm_dsigma1_sq; const float* dm_dsigma12;
    float* dL_dimg1;                            // same layout as img1
    float g_l1;                                 // fused-loss mode: + g_l1 * sign(img1 - img2), and the clamp's gradient mask
};

__global__ void __launch_bounds__(256) ssim_bwd_kernel(const SsimBwdArgs a) {
    __shared__ float s_data[3][SS_T][SS_T];
    __shared__ float s_conv[SS_T][SS_B][3];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * SS_B, y0 = blockIdx.y * SS_B;
    const int px = x0 + tx, py = y0 + ty;
    const bool in_img = px < a.W && py < a.H;
    for (int c = 0; c < a.CH; ++c) {
        for (int i = threadIdx.x; i < SS_T * SS_T; i += 256) {
            const int ly = i / SS_T, lx = i % SS_T;
            const int gy = y0 + ly - SS_HALO, gx = x0 + lx - SS_HALO;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) {
                const size_t gi = ((size_t(b) * a.CH + c) * a.H + gy) * a.W + gx;
                const float chain = a.dL_dmap ? a.dL_dmap[gi] : 1.f;
                v0 = a.dm_dmu1[gi] * chain; v1 = a.dm_dsigma1_sq[gi] * chain; v2 = a.dm_dsigma12[gi] * chain;
            }
            s_data[0][ly][lx] = v0; s_data[1][ly][lx] = v1; s_data[2][ly][lx] = v2;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SS_T * SS_B; i += 256) {
            const int ly = i / SS_B, lx = (i % SS_B) + SS_HALO;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int d = 1; d <= SS_HALO; ++d) {
                const float w = c_gauss[SS_HALO - d];
                a0 += (s_data[0][ly][lx - d] + s_data[0][ly][lx + d]) * w;
                a1 += (s_data[1][ly][lx - d] + s_data[1][ly][lx + d]) * w;
                a2 += (s_data[2][ly][lx - d] + s_data[2][ly][lx + d]) * w;
            }
            const float wc = c_gauss[SS_HALO];
            a0 += s_data[0][ly][lx] * wc; a1 += s_data[1][ly][lx] * wc; a2 += s_data[2][ly][lx] * wc;
            float* o = s_conv[ly][lx - SS_HALO];
            o[0] = a0; o[1] = a1; o[2] = a2;
        }
        __syncthreads();
        if (in_img) {
            const int ly = ty + SS_HALO;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int d = 1; d <= SS_HALO; ++d) {
                const float w = c_gauss[SS_HALO - d];
                const float* t = s_conv[ly - d][tx]; const float* u = s_conv[ly + d][tx];
                s0 += (t[0] + u[0]) * w; s1 += (t[1] + u[1]) * w; s2 += (t[2] + u[2]) * w;
            }
            const float wc = c_gauss[SS_HALO]; const float* m = s_conv[ly][tx];
            s0 += m[0] * wc; s1 += m[1] * wc; s2 += m[2] * wc;
            const float p2 = a.img2[((size_t(b) * a.CH + c) * a.H + py) * a.W + px];
            if (a.img1.hwc_clamp) {
                const size_t gi = ssim_index(a.img1, b, c, py, px, a.CH, a.H, a.W);
                const float raw = a.img1.p[gi];
                const bool clamped = a.img1.clamp != 0;
                const float p1 = clamped ? fminf(fmaxf(raw, 0.f), 1.f) : raw;
                float g = s0 + (2.f * p1) * s1 + p2 * s2;
                const float diff = p1 - p2;
                g += a.g_l1 * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)); // torch::l1_loss: sign(), 0 at 0
                a.dL_dimg1[gi] = (!clamped || (raw >= 0.f && raw <= 1.f)) ? g : 0.f; // torch::clamp passes the gradient on [0, 1]
            } else {
                const size_t gi = ((size_t(b) * a.CH + c) * a.H + py) * a.W + px;
                const float p1 = a.img1.p[gi];
                a.dL_dimg1[gi] = s0 + (2.f * p1) * s1 + p2 * s2;
            }
        }
        __syncthreads();
    }
}

static dim3 ssim_grid(int B, int H, int W) { return dim3((W + SS_B - 1) / SS_B, (H + SS_B - 1) / SS_B, B); }

} // namespace lfs

using namespace lfs;

extern "C" int lfs_fused_ssim_fwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                                  float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, lfs_stream_t stream) {
    if (B == 0 || CH == 0 || H == 0 || W == 0) return LFS_OK;
    if (!img1 || !img2 || !ssim_map) return LFS_E_INVALID;
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr)) return LFS_E_INVALID;
    SsimFwdArgs a{};
    a.H = int(H); a.W = int(W); a.CH = int(CH); a.C1 = C1; a.C2 = C2;
    a.img1 = SsimImg{img1, 0, 0, 0}; a.img2 = img2; a.ssim_map = ssim_map;
    a.dm_dmu1 = dm_dmu1; a.dm_dsigma1_sq = dm_dsigma1_sq; a.dm_dsigma12 = dm_dsigma12;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("ssim_fwd", s);
    hipLaunchKernelGGL(ssim_fwd_kernel, ssim_grid(int(B), int(H), int(W)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" int lfs_fused_ssim_bwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                                  const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                                  float* dL_dimg1, lfs_stream_t stream) {
    (void)C1; (void)C2; // carried by the reference signature, unused by its backward kernel as well
    if (B == 0 || CH == 0 || H == 0 || W == 0) return LFS_OK;
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) return LFS_E_INVALID;
    SsimBwdArgs a{};
    a.H = int(H); a.W = int(W); a.CH = int(CH);
    a.img1 = SsimImg{img1, 0, 0, 0}; a.img2 = img2; a.dL_dmap = dL_dmap;
    a.dm_dmu1 = dm_dmu1; a.dm_dsigma1_sq = dm_dsigma1_sq; a.dm_dsigma12 = dm_dsigma12; a.dL_dimg1 = dL_dimg1;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("ssim_bwd", s);
    hipLaunchKernelGGL(ssim_bwd_kernel, ssim_grid(int(B), int(H), int(W)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" size_t lfs_photometric_loss_workspace_bytes(uint32_t H, uint32_t W) { return size_t(3) * 3 * H * W * sizeof(float); }

// *loss += weight * ((1 - lambda) * mean|clamp(render) - target| + lambda * (1 - mean_valid SSIM(clamp(render), target)));
// v_render = d(that)/d(render). render / v_render HWC [H,W,3] (un-clamped rasterizer output), target CHW [3,H,W].
static int photometric_loss(uint32_t H, uint32_t W, const float* render_hwc, int render_is_chw, int clamp, const float* target_chw, float lambda_dssim,
                            float weight, float* v_render_hwc, float* loss, void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    if (H == 0 || W == 0) return LFS_OK;
    if (!render_hwc || !target_chw || !v_render_hwc || !loss || !workspace) return LFS_E_INVALID;
    if (workspace_bytes < lfs_photometric_loss_workspace_bytes(H, W)) return LFS_E_WORKSPACE;
    const int CH = 3;
    const bool crop = H > 10 && W > 10; // fused_ssim.cuh:62-66 ("valid" padding)
    const double n_map = crop ? double(H - 10) * double(W - 10) * CH : double(H) * W * CH;
    const double n_img = double(H) * W * CH;
    float* maps = (float*)workspace;
    const size_t plane = size_t(CH) * H * W;
    SsimFwdArgs f{};
    f.H = int(H); f.W = int(W); f.CH = CH; f.C1 = 0.01f * 0.01f; f.C2 = 0.03f * 0.03f;
    f.img1 = SsimImg{render_hwc, 1, render_is_chw, clamp}; f.img2 = target_chw; f.ssim_map = nullptr;
    f.dm_dmu1 = maps; f.dm_dsigma1_sq = maps + plane; f.dm_dsigma12 = maps + 2 * plane;
    f.fused = 1;
    // d loss / d ssim_map inside the crop. Reference quirk kept: when the image is too small to crop (H or W <= 10) the
    // autograd wrapper back-propagates an all-zero map (fused_ssim.cuh:88-98), i.e. the SSIM term has NO gradient there.
    f.chain = crop ? float(-double(weight) * lambda_dssim / n_map) : 0.f;
    f.crop = crop ? 1 : 0; f.loss = loss; f.loss_const = weight * lambda_dssim;
    f.w_ssim = float(-double(weight) * lambda_dssim / n_map); f.w_l1 = float(double(weight) * (1.0 - lambda_dssim) / n_img);
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("photometric_loss", s);
    hipLaunchKernelGGL(ssim_fwd_kernel, ssim_grid(1, int(H), int(W)), dim3(256), 0, s, f);
    SsimBwdArgs b{};
    b.H = int(H); b.W = int(W); b.CH = CH;
    b.img1 = SsimImg{render_hwc, 1, render_is_chw, clamp}; b.img2 = target_chw; b.dL_dmap = nullptr;
    b.dm_dmu1 = f.dm_dmu1; b.dm_dsigma1_sq = f.dm_dsigma1_sq; b.dm_dsigma12 = f.dm_dsigma12; b.dL_dimg1 = v_render_hwc;
    b.g_l1 = f.w_l1;
    hipLaunchKernelGGL(ssim_bwd_kernel, ssim_grid(1, int(H), int(W)), dim3(256), 0, s, b);
    return (int)hipGetLastError();
}

extern "C" int lfs_photometric_loss_fwd_bwd(uint32_t H, uint32_t W, const float* render_hwc, const float* target_chw, float lambda_dssim,
                                            float weight, float* v_render_hwc, float* loss, void* workspace, size_t workspace_bytes,
                                            lfs_stream_t stream) {
    return photometric_loss(H, W, render_hwc, 0, 1, target_chw, lambda_dssim, weight, v_render_hwc, loss, workspace, workspace_bytes, stream);
}

// the same loss for the fastgs rasterizer's CHW image (trainer.cpp:656-695: fast_rasterize -> compute_photometric_loss)
extern "C" int lfs_photometric_loss_chw_fwd_bwd(uint32_t H, uint32_t W, const float* render_chw, const float* target_chw, float lambda_dssim,
                                                float weight, float* v_render_chw, float* loss, void* workspace, size_t workspace_bytes,
                                                lfs_stream_t stream) {
    return photometric_loss(H, W, render_chw, 1, 0, target_chw, lambda_dssim, weight, v_render_chw, loss, workspace, workspace_bytes, stream);
}

// general form: any of the two layouts, with or without the clamp (an image that went through the bilateral grid reaches the loss un-clamped
// in either layout, trainer.cpp:662-676)
extern "C" int lfs_photometric_loss_ex_fwd_bwd(uint32_t H, uint32_t W, const float* render, uint32_t render_is_chw, uint32_t clamp_render, const float* target_chw,
                                               float lambda_dssim, float weight, float* v_render, float* loss, void* workspace, size_t workspace_bytes,
                                               lfs_stream_t stream) {
    return photometric_loss(H, W, render, render_is_chw ? 1 : 0, clamp_render ? 1 : 0, target_chw, lambda_dssim, weight, v_render, loss, workspace, workspace_bytes, stream);
}

// K7 / K8 — world-space (3DGUT) alpha compositing and its backward (replace
// gsplat::rasterize_to_pixels_from_world_3dgs_fwd / _bwd; reference:
// gsplat/RasterizeToPixelsFromWorld3DGSFwd.cu:19-279, ...Bwd.cu:16-373,
// host gsplat/Rasterization.cpp:20-261, helpers gsplat/Utils.cuh:80-194).
//
// CDNA4 design (not the reference's block-cooperative shared-memory batches):
//  * pack   : one pass turns every Gaussian into a 64-byte record - all the
//             per-Gaussian work the reference redoes per tile batch (rotmat,
//             1/scale, matrix product) happens once per Gaussian, coalesced.
//             Global shutter (round 6, LFS_REC_ROT): { U M (9), G^2, G, log2
//             opacity, rgb } with M = c S^-1 R^T Rinv and U the rotation that
//             puts g = M (o - mu) on the third axis: the distance of a ray to
//             the centre is then |g| sin(angle) = G^2 m / l from the three
//             components of q = U M d alone - no foot vector in the forward,
//             no difference of large numbers anywhere (lfs_raster_common.cuh).
//             Rolling shutters: { M (9), mu (3), ... }, per-pixel origins.
//  * raster : a wavefront owns an 8x8 pixel cell of a tile and walks the tile's
//             depth-sorted list ALONE: the list position is wave-uniform, so the
//             record arrives through the scalar unit (s_load_dwordx16 into
//             SGPRs, served by the scalar cache / L2) and feeds v_fma_f32
//             directly. No LDS staging, no __syncthreads, no 64-lane broadcast
//             reads; early-out, the alpha < 1/255 skip and the backward's
//             "behind the last contributor" skip are per 8x8 cell (ballot),
//             4x finer than the reference's 16x16 block.
//  * bwd    : with a = s w (s = alpha dL/dalpha, w the foot vector) the record's
//             gradient is dL/dM = -B M^-T, B = sum s w w^T, and dL/dg = -sum a
//             (LFS_ACC_SYM): a lane accumulates B (6), a (3), dL/dopacity,
//             dL/drgb = 13 floats; an LDS transpose (ds_write_addtid_b32 rows,
//             ds_read_b128 columns, two quad-permute DPP adds) leaves the wave
//             sums one per quad and ONE global_atomic_add_f32 instruction per
//             (wave, Gaussian) adds them to a [C*N][16] accumulator (the
//             reference: 14 shuffle reductions x 5 steps + 14 scalar atomics per
//             (warp, Gaussian)). Rolling shutters: dL/dM (9) + dL/dg (3).
//  * finish : one pass maps the accumulator through the quaternion / scale /
//             mean vjp (done per (pixel, Gaussian) in the reference);
//             dL/dscale_c = B_cc / s_c has no cancellation left in it.
// Measured and removed in round 5 (commit f684d4b, profiles/r05/lease4/ab_fwdzero_off.txt): the forward kernel clearing the backward's accumulator rows on the side (two or three
// 16-byte stores per lane at kernel start, instead of the 64 MB hipMemsetAsync in front of the backward: 9 - 10 us): raster_fwd 0.234 -> 0.247 ms, step +0.015 - 0.025 ms - stores
// issued by a VALU-bound kernel are not free. Round 6 (profiles/r06/lease15_ab_tail_acc_clear.txt): the same 64 MB cleared by the step's PROJECTION kernel (four 16-byte stores per lane at its
// top): projection 0.064 -> 0.081 ms, step +0.007 ms against the memset's 0.0115 ms + launch gap - that kernel moves 204 MB in 64 us and has no memory slack either: removed.
// Measured and removed in round 3 (kernels in git history up to e34272a, numbers under profiles/): 16x8 "wide" cells with two pixels per lane
// (profiles/r01/raster_wide_cells_ab.json: bwd 0.84 vs 0.69 ms), quadrant-row kernels with DPP-broadcast records (profiles/r02/raster_rows_vs_default_pmc.txt:
// 1.17x the VALU instructions), SH colours + record packing in one kernel (profiles/r02/fuse_front_ab.txt: no gain).
#include "lfs_camera.cuh"
#include "lfs_prof.h"
#ifndef LFS_EMULATE
#include <hip/hip_ext.h>
#endif
#ifndef LFS_PROF_EXT_LAUNCH
#define LFS_PROF_EXT_LAUNCH 1   // 0: two hipEventRecord around the backward's launch (rounds 1 - 6)
#endif
#include "lfs_raster_common.cuh"
#include "lfs_cull_conic.cuh"
#include "lfs_raster_pack.cuh"
#include "lfs_adam.cuh"
#include "lfs_sh.cuh"
#include "lfs_step_internal.h"

// LFS_BWD_REORTH (default 1 since round 5; -DLFS_BWD_REORTH=0 = the rounds 1 - 4 backward, kept for A/B: tools/build_variant.py noreorth raster.hip -DLFS_BWD_REORTH=0): the backward
// re-orthogonalises the foot vector against the ray direction before it is used in a gradient - K8's gradients for FLAT Gaussians (tools/aniso_probe.py).
#ifndef LFS_BWD_REORTH
#define LFS_BWD_REORTH 1
#endif
#if LFS_SEL_E64 && !LFS_REC_LOG2
#error "LFS_SEL_E64 is written for the LFS_REC_LOG2 records"
#endif
#ifndef LFS_FWD_MARK
#define LFS_FWD_MARK 0   // 1: the forward marks the cell-list entries nothing composited and the backward skips them. Measured (round 6, profiles/r06/lease18_fwd_marks_projfast_ab.txt,
#endif                   // same box, 4 x 200 steps): raster_fwd 0.233 -> 0.256 ms, raster_bwd 0.4856 -> 0.4845 ms, 734 -> 722 img/s: off.
#ifndef LFS_FINISH_LDS_ROWS
#define LFS_FINISH_LDS_ROWS 1 // (round 3, same box: finish_adam 0.106 / 0.102 -> 0.102 / 0.097 ms; 0 = four 16-byte loads per lane at a 64-byte stride)
#endif
// Host build on the wavefront emulator (tests/emul) only: wave-evaluation counters [fwd, fwd that composited, bwd, bwd that accumulated]
#ifdef LFS_EMULATE
extern "C" { __attribute__((visibility("default"))) unsigned long long lfs_emul_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0}; }
#define LFS_EMUL_COUNT(i) do { if ((threadIdx.x & 63) == 0) ++lfs_emul_counters[i]; } while (0)
// lane utilisation of an accumulated backward evaluation: [4] += live lanes, [5] += 8x4 half cells (rows 0-3 / 4-7) with a live lane, [6] += 4x4 quarters with one
#define LFS_EMUL_LANES(m) do { const unsigned long long m_ = (m); if ((threadIdx.x & 63) == 0) { lfs_emul_counters[4] += __builtin_popcountll(m_); \
    lfs_emul_counters[5] += ((m_ & 0xffffffffull) != 0) + ((m_ >> 32) != 0); \
    for (int q_ = 0; q_ < 4; ++q_) { const unsigned long long qm_ = (0x0f0f0f0full << ((q_ & 1) * 4)) << ((q_ >> 1) * 32); lfs_emul_counters[6] += (m_ & qm_) != 0; } } } while (0)
#else
#define LFS_EMUL_COUNT(i) do { } while (0)
#define LFS_EMUL_LANES(m) do { } while (0)
#endif

namespace lfs {

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
constexpr uint32_t LOSS_SLOTS = 256; // fused MSE: the wavefronts' partial sums are spread over this many addresses (one hot address costs ~0.08 ms)
struct RasterWs { CamDev* cams; GaussRec* recs; float* acc; CullRec* cull; int32_t* cell_count; int2* cell_list; unsigned long long* det64; size_t bytes; };
// cells = C * tiles * (tile_size/8)^2 ; the compacted per-cell lists hold at most (tile_size/8)^2 * n_isects entries
static RasterWs raster_ws(void* base, uint32_t C, uint32_t N, uint64_t cells, uint64_t cell_entries, bool det = false) {
    RasterWs w; char* p = (char*)base; size_t o = 0;
    w.cams = (CamDev*)(p + o); o += align256(sizeof(CamDev) * C);
    w.recs = (GaussRec*)(p + o); o += align256(sizeof(GaussRec) * size_t(C) * N);
    w.acc = (float*)(p + o); o += align256(sizeof(float) * (ACC_STRIDE * size_t(C) * N + LOSS_SLOTS)); // + the fused-loss partial sums
    w.cull = (CullRec*)(p + o); o += align256(sizeof(CullRec) * size_t(C) * N);
    w.cell_count = (int32_t*)(p + o); o += align256(sizeof(int32_t) * cells);
    w.cell_list = (int2*)(p + o); o += align256(sizeof(int2) * cell_entries);
    w.det64 = nullptr;
    if (det) { w.det64 = (unsigned long long*)(p + o); o += align256(sizeof(unsigned long long) * ACC_STRIDE * size_t(C) * N); } // deterministic backward (debug bit 4)
    w.bytes = o;
    return w;
}

__global__ void cam_prep_kernel(const lfs_cameras cams, CamDev* __restrict__ out) {
    for (uint32_t c = threadIdx.x; c < cams.C; c += blockDim.x) {
        CamDev cd;
        cam_init(cd, cams, c);
        out[c] = cd;
    }
}

// ---------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------
template <bool UNIFORM_ORIGIN>
__global__ void __launch_bounds__(256) raster_pack_kernel(
    const uint32_t C, const uint32_t N, const uint32_t channels,
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const CamDev* __restrict__ cams, GaussRec* __restrict__ recs, CullRec* __restrict__ cull) {
    const size_t idx = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= size_t(C) * N) return;
    const uint32_t cid = uint32_t(idx / N), gid = uint32_t(idx % N);
    const f3 mu{means[3 * gid], means[3 * gid + 1], means[3 * gid + 2]};
    const float4 q = reinterpret_cast<const float4*>(quats)[gid];
    const float sc[3] = {scales[3 * gid], scales[3 * gid + 1], scales[3 * gid + 2]};
    const float* cp = colors + idx * channels;
    GaussRec rec;
    CullRec cr;
    pack_gaussian<UNIFORM_ORIGIN>(cams[cid], mu, q, sc, opacities[idx], cp[0], channels > 1 ? cp[1] : 0.f, channels > 2 ? cp[2] : 0.f, rec, cr);
    recs[idx] = rec;
    cull[idx] = cr;
}

// ---------------------------------------------------------------------------
// cull: per 8x8 cell, compact the tile's depth-sorted list down to the entries that CAN reach the
// 1/255 alpha threshold on at least one of the cell's rays (conservative: never drops a contributor, so
// fwd/bwd results are exactly those of walking the full tile list). Lane = list entry; the cell's box of rays in normalised
// camera coordinates is wave-uniform and tested against the entry's silhouette conic (lfs_cull_conic.cuh). Output per cell: count + (gaussian, list index)
// pairs in list order, stored in the cell's slice of a [cells_per_tile * n_isects] array.
// ---------------------------------------------------------------------------

#ifndef LFS_CULL_DEPTH
#define LFS_CULL_DEPTH 1   // batches of look-ahead per thread in raster_cull_kernel; 2 and 4 measured: no gain (0.072 -> 0.073 / 0.078 ms on SYN-B) - the kernel is
#endif                     // bound by the gather throughput (L2 / texture-address unit), not by the latency of a workgroup's dependent loads
template <bool UNIFORM_ORIGIN>
__global__ void __launch_bounds__(256) raster_cull_kernel(
    const uint32_t C, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block, const uint32_t cull_enabled,
    const CamDev* __restrict__ cams, const CullRec* __restrict__ cull, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ ids, const int32_t n_isects,
    int32_t* __restrict__ cell_count, int2* __restrict__ cell_list) {
    // The (up to 4) cells of a workgroup belong to the same tile and walk the same list: the workgroup gathers each
    // batch of 64 * waves entries ONCE (one entry per thread: id + 32-B culling record, a dependent random gather that
    // would otherwise be repeated by every cell and keep the texture-address unit busy) and hands it over through LDS;
    // every wave then tests the whole batch against its own cell. Double-buffered: one barrier per batch.
    __shared__ float4 s_a[2][256], s_b[2][256];
    __shared__ int32_t s_g[2][256];
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return; // (uniform per workgroup)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wpt = (tile_size >> 3) * (tile_size >> 3);
    const size_t cell = size_t(cc.tile_global) * wpt + cc.wl;
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1 && n_isects >= 0) ? n_isects : offsets[cc.tile_global + 1]; // n_isects < 0: offsets has T + 1 entries (guarded step)
    const bool tile_masked = masks != nullptr && !masks[cc.tile_global];
    if (tile_masked || end <= start) { // uniform per workgroup
        if (lane == 0) cell_count[cell] = 0;
        return;
    }

    // cell bounds in normalised camera coordinates (x/z, y/z) over the rays that can composite at all
    const CamDev& cam = cams[cc.cid];
    const float big = 3.0e38f;
    bool active = false, behind = false;
    float tu_min = big, tu_max = -big, tv_min = big, tv_max = -big;
    auto probe = [&](uint32_t j, uint32_t i) {
        f3 ro, rd;
        const bool ok = cam_pixel_ray(cam, f2{float(j) + 0.5f, float(i) + 0.5f}, ro, rd);
        const bool act = i < H && j < W && ok;
        active = active || act;
        if (UNIFORM_ORIGIN) {
            const f3 cd = mul_t(cam.Rinv, rd); // back to camera space
            const bool front = cd.z > 0.f;
            behind = behind || (act && !front);
            const float iz = front ? 1.f / cd.z : 0.f;
            const float tu = cd.x * iz, tv = cd.y * iz;
            tu_min = fminf(tu_min, act ? tu : big); tu_max = fmaxf(tu_max, act ? tu : -big);
            tv_min = fminf(tv_min, act ? tv : big); tv_max = fmaxf(tv_max, act ? tv : -big);
        }
    };
    probe(cc.j, cc.i);
    const bool cell_live = __ballot(active) != 0ull; // a dead cell still helps with the loads and the barriers
    bool can_cull = UNIFORM_ORIGIN && cull_enabled != 0;
    float tu_lo = 0.f, tu_hi = 0.f, tv_lo = 0.f, tv_hi = 0.f;
    if (UNIFORM_ORIGIN) {
        can_cull = can_cull && (__ballot(behind) == 0ull);
        const float mu_ = 0.25f / cam.fx, mv_ = 0.25f / cam.fy; // numerical safety margin: a quarter pixel
        tu_lo = wave_min(tu_min) - mu_;
        tu_hi = wave_max(tu_max) + mu_;
        tv_lo = wave_min(tv_min) - mv_;
        tv_hi = wave_max(tv_max) + mv_;
        can_cull = can_cull && (tu_hi - tu_lo < 1e30f) && (tv_hi - tv_lo < 1e30f);
    }
    tu_lo = uniform_f(tu_lo); tu_hi = uniform_f(tu_hi); tv_lo = uniform_f(tv_lo); tv_hi = uniform_f(tv_hi);
    const bool need_recs = UNIFORM_ORIGIN && cull_enabled != 0; // workgroup-uniform (can_cull is per cell)

    int2* __restrict__ out = cell_list + (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    int32_t count = 0;
    const int32_t E = int32_t(blockDim.x);          // entries per batch
    const int32_t nsub = E >> 6;                    // = waves in the workgroup
    auto fetch = [&](int32_t base, int32_t& g, CullRec& cr) {
        const int32_t i = base + int32_t(threadIdx.x);
        g = i < end ? ids[i] : 0;
        if (need_recs) cr = cull[g];
    };
    // DEPTH batches are in flight per thread (id load -> dependent record gather: two memory round trips each)
    constexpr int DEPTH = LFS_CULL_DEPTH;
    int32_t g_reg[DEPTH]; CullRec cr_reg[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        cr_reg[d].a = make_float4(0.f, 0.f, 0.f, 0.f); cr_reg[d].b = cr_reg[d].a; g_reg[d] = 0;
        if (start + d * E < end) fetch(start + d * E, g_reg[d], cr_reg[d]);
    }
    int buf = 0;
    for (int32_t base0 = start; base0 < end; base0 += DEPTH * E) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int32_t base = base0 + d * E;
            if (base >= end) break; // uniform
            s_g[buf][threadIdx.x] = g_reg[d];
            if (need_recs) { s_a[buf][threadIdx.x] = cr_reg[d].a; s_b[buf][threadIdx.x] = cr_reg[d].b; }
            __syncthreads();
            if (base + DEPTH * E < end) fetch(base + DEPTH * E, g_reg[d], cr_reg[d]); // in flight during the tests below
            if (cell_live) {
                for (int32_t sub = 0; sub < nsub; ++sub) {
                    const int32_t slot = (sub << 6) + int32_t(lane);
                    const int32_t my_idx = base + slot;
                    if (base + (sub << 6) >= end) break; // uniform
                    const bool valid = my_idx < end;
                    const int32_t my_g = s_g[buf][slot];
                    bool hit = valid;
                    if (can_cull) {
                        const float4 a = s_a[buf][slot], b = s_b[buf][slot];
                        hit = valid && !conic_culled(ConicRec{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, tu_lo, tu_hi, tv_lo, tv_hi);
                    }
                    const uint64_t m = __ballot(hit);
                    if (hit) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                        out[count + int32_t(rank)] = make_int2(my_g, my_idx);
                    }
                    count += __popcll(m);
                }
            }
            buf ^= 1;
        }
    }
    if (lane == 0) cell_count[cell] = count;
}

// Ray modes (template parameter of fwd / bwd):
//   0 rolling shutter : world-space ray (ro, rd) per pixel, record = {M, mu}
//   1 global shutter  : camera-space direction d (any camera model), record = {M Rinv, M (o - mu)}
// (A third mode with d = (u, v, 1) for pinholes was tried: the three multiplications it saves come back as v_mov,
//  because a VOP3 instruction on gfx9 can read only one SGPR and fma(M01, v, M02) needs two.)
constexpr int RAY_ROLLING = 0, RAY_GLOBAL = 1;

// Distance of the Gaussian centre to the ray line in the Gaussian's normalised frame. With q = M d (un-normalised),
// t = (gro . q) / |q|^2 and the foot vector w = gro - t q:  |w| equals the reference's |normalize(q) x gro|, and the
// backward collapses to dL/dgro = -s w, dL/dq = t s w (s = vis * dL/dvis): no normalisation, no cross products.
struct RayEval { f3 om, w, q; float t, vis, rl; }; // LFS_REC_LOG2: w = c w_true and `vis` is alpha_raw = opac * vis_true (see lfs_raster_common.cuh)
template <int MODE>
LFS_DI void ray_eval(const GaussRec& rec, const f3& ro, const f3& d, RayEval& e) {
    e.om = {0.f, 0.f, 0.f};
#if LFS_REC_ROT
    if (MODE == RAY_GLOBAL) { // the record lives in the frame in which g = (0, 0, G) (lfs_raster_common.cuh, LFS_REC_ROT): |w|^2 = G^2 m / l, no difference of large numbers anywhere
#if LFS_REC_PKQ
        v2f qxy = v2f{rec.r0.x, rec.r0.y} * v2f{d.x, d.x};
        qxy = __builtin_elementwise_fma(v2f{rec.r0.z, rec.r0.w}, v2f{d.y, d.y}, qxy);
        qxy = __builtin_elementwise_fma(v2f{rec.r1.x, rec.r1.y}, v2f{d.z, d.z}, qxy);
        const f3 q{qxy.x, qxy.y, fma3(rec.r2.x, d.x, rec.r2.y, d.y, rec.r2.z, d.z)};
        const float rec_k = rec.r1.z;
#else
        const f3 q{fma3(rec.r0.x, d.x, rec.r0.y, d.y, rec.r0.z, d.z),
                   fma3(rec.r1.x, d.x, rec.r1.y, d.y, rec.r1.z, d.z),
                   fma3(rec.r2.x, d.x, rec.r2.y, d.y, rec.r2.z, d.z)};
        const float rec_k = rec.r0.w;
#endif
        const float m = __builtin_fmaf(q.y, q.y, q.x * q.x);
        const float l = __builtin_fmaf(q.z, q.z, m);
        const float rl = fast_rcp(l);      // l == 0 (inactive lane: d = 0; degenerate record): inf, and m = q.z = 0 - the two products below are v_mul_legacy_f32 (0 * inf = 0): u = 0, t = 0, w = 0
        const float u = mul_zero(m, rl);   // sin^2 of the angle between the ray and the direction to the centre
        e.vis = __builtin_amdgcn_exp2f(__builtin_fmaf(-rec_k, u, rec.r3.x));
        e.t = rec.r2.w * mul_zero(q.z, rl);
#ifndef LFS_EMULATE
        asm volatile("" ::"s"(rec.r1.w), "s"(rec.r2.w)); // (no instruction: the unused fields of the record - r1.w, and G in the forward - stay "used", so the record still arrives as ONE s_load_dwordx16; without it the compiler splits the load into two to four)
#endif
        e.q = q; e.rl = rl;
        e.w = {-e.t * q.x, -e.t * q.y, rec.r2.w * u};
        return;
    }
#endif
    f3 gro;
    if (MODE != RAY_ROLLING) gro = {rec.r0.w, rec.r1.w, rec.r2.w};
    else {
        e.om = {ro.x - rec.r0.w, ro.y - rec.r1.w, ro.z - rec.r2.w};
        gro = {fma3(rec.r0.x, e.om.x, rec.r0.y, e.om.y, rec.r0.z, e.om.z),
               fma3(rec.r1.x, e.om.x, rec.r1.y, e.om.y, rec.r1.z, e.om.z),
               fma3(rec.r2.x, e.om.x, rec.r2.y, e.om.y, rec.r2.z, e.om.z)};
    }
    const f3 q{fma3(rec.r0.x, d.x, rec.r0.y, d.y, rec.r0.z, d.z),
               fma3(rec.r1.x, d.x, rec.r1.y, d.y, rec.r1.z, d.z),
               fma3(rec.r2.x, d.x, rec.r2.y, d.y, rec.r2.z, d.z)};
    const float l = fma3(q.x, q.x, q.y, q.y, q.z, q.z);
#if LFS_REC_LOG2
    // l == 0 (inactive lane: d = 0; degenerate record: M = 0): q = 0 as well, so t = 0 * FLT_MAX = 0 and w = gro - one v_min instead of compare + select
    const float rl = fminf(fast_rcp(l), 3.402823466e38f);
    e.t = fma3(gro.x, q.x, gro.y, q.y, gro.z, q.z) * rl;
    e.q = q; e.rl = rl;
    e.w = {__builtin_fmaf(-e.t, q.x, gro.x), __builtin_fmaf(-e.t, q.y, gro.y), __builtin_fmaf(-e.t, q.z, gro.z)};
    e.vis = __builtin_amdgcn_exp2f(__builtin_fmaf(-e.w.z, e.w.z, __builtin_fmaf(-e.w.y, e.w.y, __builtin_fmaf(-e.w.x, e.w.x, rec.r3.x))));
#else
    const float rl = l > 0.f ? fast_rcp(l) : 0.f; // l == 0: no direction (inactive lane / degenerate record), w = gro
    e.t = fma3(gro.x, q.x, gro.y, q.y, gro.z, q.z) * rl;
    e.q = q; e.rl = rl;
    e.w = {__builtin_fmaf(-e.t, q.x, gro.x), __builtin_fmaf(-e.t, q.y, gro.y), __builtin_fmaf(-e.t, q.z, gro.z)};
    // exp(-0.5 |w|^2) as one exp2: -0.5 * log2(e) = -0.72134752
    e.vis = __builtin_amdgcn_exp2f(-0.72134752044448170f * fma3(e.w.x, e.w.x, e.w.y, e.w.y, e.w.z, e.w.z));
#endif
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// this lane's ray in the form the mode wants; false when the pixel has no ray
template <int MODE>
LFS_DI bool lane_ray(const CamDev& cam, const uint32_t j, const uint32_t i, f3& ro, f3& d) {
    const f2 ip{float(j) + 0.5f, float(i) + 0.5f};
    if (MODE == RAY_ROLLING) return cam_pixel_ray(cam, ip, ro, d);
    ro = cam.origin;
    f3 cd;
    const bool ok = cam_unproject(cam, ip, cd);
    d = ok ? cd : f3{0.f, 0.f, 0.f};
    return ok;
}

template <int CDIM, int MODE>
__global__ void __launch_bounds__(256) raster_fwd_kernel(
    const uint32_t C, const uint32_t N, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block,
    const CamDev* __restrict__ cams, const GaussRec* __restrict__ recs, const float* __restrict__ colors,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ cell_count, const int2* __restrict__ cell_list, const int32_t n_isects,
    float* __restrict__ render_colors, float* __restrict__ render_alphas, int32_t* __restrict__ last_ids, int32_t* __restrict__ cell_marks) {
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return;
    const uint32_t cid = cc.cid;
    const bool inside = cc.i < H && cc.j < W;
    const size_t pix_id = (size_t(cid) * H + cc.i) * W + cc.j;
    const float* bg = backgrounds ? backgrounds + cid * CDIM : nullptr;

    if (masks != nullptr && !masks[cc.tile_global]) { // Fwd.cu:141-150 (alpha / last id written as 0 instead of left undefined)
        if (inside) {
#pragma unroll
            for (int k = 0; k < CDIM; ++k) render_colors[pix_id * CDIM + k] = bg ? bg[k] : 0.f;
            render_alphas[pix_id] = 0.f;
            last_ids[pix_id] = 0;
        }
        return;
    }

    const CamDev& cam = cams[cid];
    f3 ro, rd;
    const bool ray_ok = lane_ray<MODE>(cam, cc.j, cc.i, ro, rd);
    // "done" is carried as the lane's alpha threshold: 1/255 while the pixel is live, +inf once it has terminated (or never
    // had a ray). One compare then answers both "not done" and "alpha >= 1/255", and the per-lane state stays out of the
    // boolean VGPR juggling the compiler otherwise emits for a loop-carried bool (6 VALU per evaluation, measured in the ISA).
    const float INF = __builtin_inff();
    float thr = (inside && ray_ok) ? (1.f / 255.f) : INF;

    const uint32_t wpt = (tile_size >> 3) * (tile_size >> 3);
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1 && n_isects >= 0) ? n_isects : offsets[cc.tile_global + 1]; // n_isects < 0: offsets has T + 1 entries (guarded step)
    const int2* __restrict__ cl = cell_list + (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    const int32_t cnt = cell_count[size_t(cc.tile_global) * wpt + cc.wl];
#if LFS_FWD_MARK
    // Round 6: an entry whose alpha stays below 1/255 on every live ray of this cell (the culling in front is conservative) cannot pass the backward's test either (a lane valid there composited here: e.y <= its last contributor and alpha >= 1/255, the same
    // bits). The forward says so in the list itself - the sign bit of the entry's Gaussian index, one 4-byte store by one lane - and the backward skips the entry before
    // it evaluates anything (14 % of its evaluations on the dense SYN-B window of profiles/r05/quarter_histogram.txt: 31 VALU instructions each). The record address of
    // a marked entry is unchanged: the walker forms it as uint32(index) << 6 and the bit falls off the top. Entries the forward never reaches lie behind the last
    // contributor, where the backward does not walk.
    // cell_marks IS the cell list (host: the same pointer), handed over as a second __restrict__ argument: a store through cell_list itself makes the list "written in
    // this kernel" for the compiler, and the walker's entry loads stop being scalar loads (measured: raster_fwd 0.238 -> 0.399 ms). A position is never read again
    // once it is marked (the walker runs ahead of the evaluation), so the two views of the memory never meet.
    int32_t* const cl_mark = cell_marks + 2 * (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    const bool marker = (threadIdx.x & 63u) == 0u;
    int32_t pos = 0;   // (uniform) the evaluations run over positions 0, 1, 2, ... of the list
    auto mark = [&](const int32_t p, const int32_t g) { if (marker) cl_mark[2 * p] = g | int32_t(0x80000000u); };
#endif

    float T = 1.f;
    float pix[CDIM];
#pragma unroll
    for (int k = 0; k < CDIM; ++k) pix[k] = 0.f;
    int32_t cur_idx = 0;

    // One evaluation of a record against this lane's ray (wave-uniform record in SGPRs).
    auto eval = [&](const GaussRec& rec, const int2 e) {
        RayEval re;
        ray_eval<MODE>(rec, ro, rd, re);
        const float alpha = fminf(0.999f, LFS_REC_LOG2 ? re.vis : rec.r3.x * re.vis);
#if LFS_SEL_E64
        // lane-mask form (lfs_raster_common.cuh): the same compares and selects, no VCC / EXEC round trips. A lane that does not composite adds fma(c, 0, pix) = pix.
        const lmask_t pass = mask_nlt_f32(alpha, thr); // live pixel and alpha >= 1/255 (a NaN alpha passes, as in the reference's `if (alpha < 1/255) continue`)
        LFS_EMUL_COUNT(0);
#if LFS_FWD_MARK
        const int32_t my_pos = pos; pos += 1;
        if (pass == 0ull) { mark(my_pos, e.x); return; }
#else
        if (pass == 0ull) return;
#endif
        LFS_EMUL_COUNT(1);
        const float next_T = T * (1.f - alpha);
        const lmask_t fin = pass & mask_le_f32(next_T, 1e-4f); // the terminating Gaussian is not composited
        const lmask_t contrib = pass & ~fin;
        const float vis = sel_mask(contrib, alpha * T, 0.f);
        if (CDIM <= 3) {
            pix[0] = __builtin_fmaf(rec.r3.y, vis, pix[0]);
            if (CDIM > 1) pix[1] = __builtin_fmaf(rec.r3.z, vis, pix[1]);
            if (CDIM > 2) pix[2] = __builtin_fmaf(rec.r3.w, vis, pix[2]);
        } else {
            const float* cp = colors + size_t(e.x) * CDIM;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) pix[k] = __builtin_fmaf(cp[k], vis, pix[k]);
        }
        cur_idx = sel_mask_i32(contrib, e.y, cur_idx);
        T = sel_mask(contrib, next_T, T);
        thr = sel_mask(fin, INF, thr);
#else
        const bool pass = !(alpha < thr); // live pixel and alpha >= 1/255 (a NaN alpha passes, as in the reference's `if (alpha < 1/255) continue`)
        LFS_EMUL_COUNT(0);
#if LFS_FWD_MARK
        const int32_t my_pos = pos; pos += 1;
        if (__ballot(pass) == 0ull) { mark(my_pos, e.x); return; }
#else
        if (__ballot(pass) == 0ull) return;
#endif
        LFS_EMUL_COUNT(1);
        const float next_T = T * (1.f - alpha);
        const bool fin = pass && next_T <= 1e-4f; // the terminating Gaussian is not composited
        const bool contrib = pass && !fin;
        const float vis = alpha * T;
        if (contrib) {
            if (CDIM <= 3) {
                pix[0] = __builtin_fmaf(rec.r3.y, vis, pix[0]);
                if (CDIM > 1) pix[1] = __builtin_fmaf(rec.r3.z, vis, pix[1]);
                if (CDIM > 2) pix[2] = __builtin_fmaf(rec.r3.w, vis, pix[2]);
            } else {
                const float* cp = colors + size_t(e.x) * CDIM;
#pragma unroll
                for (int k = 0; k < CDIM; ++k) pix[k] = __builtin_fmaf(cp[k], vis, pix[k]);
            }
            cur_idx = e.y;
            T = next_T;
        }
        thr = fin ? INF : thr;
#endif
    };
    walk_cell_list<1>(cl, recs, 0, cnt, eval, [&]() { return __ballot(thr < INF) != 0ull; });

    if (inside) {
        render_alphas[pix_id] = 1.f - T;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) render_colors[pix_id * CDIM + k] = bg ? pix[k] + T * bg[k] : pix[k];
        last_ids[pix_id] = cur_idx;
    }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// Optional fused loss (extension, CDIM = 3, one camera): the kernel derives dL/d(render) = d/d(render) of
// weight * mean((clamp(render, 0, 1) - target)^2) from the forward image and the CHW target itself and adds the loss to *loss -
// the arithmetic of l2_fused.hip's mse_loss_kernel per pixel, so v_render_colors is never materialised and the loss kernel's
// pass over the image disappears.
struct MseFuse { const float* render; const float* target; float scale; float* loss; unsigned long long* det64; };

#ifndef LFS_RASTER_WAVE_BLOCKS
#define LFS_RASTER_WAVE_BLOCKS 1   // fwd / bwd launched with ONE wavefront per workgroup (wave_geom below). Same-box A/B x3 (profiles/r03/raster_wave_blocks_ab.txt):
                                   // raster_bwd 0.532 - 0.537 -> 0.514 - 0.526 ms, raster_fwd 0.241 - 0.247 -> 0.237 - 0.241 ms; 0 = the tile's four cells as one workgroup
#endif
#ifndef LFS_BWD_LDS_REDUCE
#define LFS_BWD_LDS_REDUCE 1   // the 16-value wave reduction through an LDS transpose instead of register swaps (lfs_raster_common.cuh). Measured on SYN-B, same box,
#endif                         // 3 pairs (profiles/r03/raster_bwd_lds_reduce_ab.txt): raster_bwd 0.621 - 0.624 -> 0.534 - 0.541 ms; 0 = the register transpose of rounds 1 - 2
#ifndef LFS_BWD_ALPHA0
#define LFS_BWD_ALPHA0 1   // (round 6) invalid lanes of a backward evaluation carry alpha = 0 instead of selects on T and fac, and the running tail - B is ONE variable: -2 VALU per evaluation
#endif
#if LFS_BWD_ALPHA0 && LFS_SEL_E64
#error "LFS_BWD_ALPHA0 is written for the bool form of the conditions"
#endif
#ifndef LFS_BWD_PK
#define LFS_BWD_PK 1   // the backward's gradient products and the first two levels of its 16-value reduction on register pairs (v_pk_mul_f32 / v_pk_add_f32): 47 fewer
#endif                 // VALU instructions in the kernel (-12 per evaluation), bit-identical sums; measured 0.621 - 0.631 -> 0.611 - 0.617 ms (same box, 3 pairs)
template <int CDIM, int MODE, bool LOSS = false, int ACC = 0>
__global__ void __launch_bounds__(256) raster_bwd_kernel(
    const uint32_t C, const uint32_t N, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block,
    const CamDev* __restrict__ cams, const GaussRec* __restrict__ recs, const float* __restrict__ colors,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ cell_count, const int2* __restrict__ cell_list, const int32_t n_isects,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ acc, float* __restrict__ v_colors_extra, const MseFuse mse = MseFuse{}) {
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
#if LFS_BWD_LDS_REDUCE
    constexpr int RED_BLOCK = (LFS_RED_QUAD_ASM && RED_QUAD_SCRATCH_FLOATS > RED_SCRATCH_FLOATS) ? RED_QUAD_SCRATCH_FLOATS : RED_SCRATCH_FLOATS;
    __shared__ __attribute__((aligned(16))) float s_red[(LFS_RASTER_WAVE_BLOCKS ? 1 : 4) * RED_BLOCK]; // one transpose block per wavefront (wave_sum16_atomic_lds / _quad)
    float* const red_scratch = s_red + (threadIdx.x >> 6) * RED_BLOCK;
#if LFS_RED_QUAD_ASM
    const uint32_t red_base = __builtin_amdgcn_readfirstlane(uint32_t(reinterpret_cast<uintptr_t>(red_scratch)));   // LDS byte address of the block (low half of the flat address)
#if LFS_RED_M0_ONCE
    asm volatile("s_mov_b32 m0, %0" ::"s"(red_base));   // the one write of M0 in this kernel (lfs_raster_common.cuh, LFS_RED_M0_ONCE)
#endif
    const float4* const red_rd = reinterpret_cast<const float4*>(red_scratch + ((threadIdx.x & 63u) >> 2) * RED_QROW + 4u * (threadIdx.x & 3u));
    constexpr bool RED_SKIP = LFS_ACC_SYM && MODE == RAY_GLOBAL && CDIM == 3;   // (slots 9 .. 11 of the LFS_ACC_SYM row are empty)
    const bool red_atomic_lane = (threadIdx.x & 3u) == 0u && !(RED_SKIP && ((threadIdx.x & 63u) >> 2) >= 9u && ((threadIdx.x & 63u) >> 2) <= 11u);
#if LFS_RED_BUF_ATOMIC
    const RedBuf red_buf = red_buf_make(acc, uint64_t(C) * N, threadIdx.x & 63u, red_atomic_lane);   // (ACC == 0: the totals leave through a buffer atomic, lfs_raster_common.cuh)
#endif
#endif
#endif
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return;
    const uint32_t cid = cc.cid;
    if (masks != nullptr && !masks[cc.tile_global]) return; // masked tiles composited nothing
    const uint32_t lane = threadIdx.x & 63;
    const bool inside = cc.i < H && cc.j < W;
    const size_t pix_id = (size_t(cid) * H + cc.i) * W + cc.j;
    const float* bg = backgrounds ? backgrounds + cid * CDIM : nullptr;

    const CamDev& cam = cams[cid];
    f3 ro, rd;
    const bool ray_ok = lane_ray<MODE>(cam, cc.j, cc.i, ro, rd);
    const bool active = inside && ray_ok;

    const uint32_t wpt = (tile_size >> 3) * (tile_size >> 3);
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1 && n_isects >= 0) ? n_isects : offsets[cc.tile_global + 1]; // n_isects < 0: offsets has T + 1 entries (guarded step)
    const int2* __restrict__ cl = cell_list + (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    const int32_t cnt = cell_count[size_t(cc.tile_global) * wpt + cc.wl];

    float T_final = 1.f, v_ra = 0.f;
    int32_t bin_final = -1; // Bwd.cu:183 uses 0 for inactive pixels; -1 keeps them out of entry 0 as well
    float vc[CDIM], Bsum = 0.f;
#pragma unroll
    for (int k = 0; k < CDIM; ++k) vc[k] = 0.f;
    if (active) {
        T_final = 1.f - render_alphas[pix_id];
        bin_final = last_ids[pix_id];
        v_ra = v_render_alphas ? v_render_alphas[pix_id] : 0.f;
        if (!LOSS) {
#pragma unroll
            for (int k = 0; k < CDIM; ++k) vc[k] = v_render_colors[pix_id * CDIM + k];
        }
    }
    if (LOSS) { // every pixel of the image belongs to exactly one lane of one wavefront
        float lsum = 0.f;
        if (inside) {
            const size_t P = size_t(H) * W;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) {
                const float x = mse.render[pix_id * CDIM + k];
                const float d = fminf(fmaxf(x, 0.f), 1.f) - mse.target[size_t(k) * P + pix_id];
                lsum += d * d;
                const float g = (x >= 0.f && x <= 1.f) ? 2.f * d * mse.scale : 0.f;
                if (active) vc[k] = g;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m, 64);
        if (ACC != 1 && lane == 0 && lsum != 0.f) unsafeAtomicAdd(mse.loss + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (LOSS_SLOTS - 1)), lsum * mse.scale);
    }
    float T = T_final;
    // T_final * (v_alpha_out - bg . v_color_out): the transmittance-tail term of d/d(alpha)
    float tail = v_ra;
    if (bg) {
        float bd = 0.f;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) bd += bg[k] * vc[k];
        tail -= bd;
    }
    tail *= T_final;

    // wave max of bin_final (uniform): nothing behind it can matter for this cell
    int32_t wmax = bin_final;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wmax = max(wmax, __shfl_xor(wmax, m, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    // number of cell-list entries whose list index is <= wmax (entries are in ascending list order)
    int32_t lo = 0, hi = cnt;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (cl[mid].y <= wmax) lo = mid + 1; else hi = mid;
    }
    const int32_t n_walk = lo;
    if (n_walk <= 0) return;

    auto eval = [&](const GaussRec& rec, const int2 e) {
#if LFS_FWD_MARK
        if (e.x < 0) return;   // (uniform) marked by the forward: no pixel of this cell composited the entry - nothing to accumulate (see raster_fwd_kernel)
#endif
        RayEval re;
        ray_eval<MODE>(rec, ro, rd, re);
#if LFS_BWD_REORTH
        if (!(LFS_REC_ROT && MODE == RAY_GLOBAL)) // (the rotated records of LFS_REC_ROT form w without a difference: nothing to take out)
        {   // w = gro - t q is the difference of two vectors of length |o - mu| / s_min: for a FLAT Gaussian (one scale 20 - 100 x below the others, 5 units away: 1e4) its
            // component ALONG q carries an absolute rounding error of ulp(1e4) ~ 1e-3 - harmless in |w|^2 (alpha above is computed from the un-corrected w, bit-identical
            // to the forward), but dL/dgro = -s w is multiplied by 1 / s_min again in the finish pass. w is orthogonal to q by construction: one Gram-Schmidt step takes
            // the spurious component out (w . q is a product with the LARGE q, so it is resolved exactly where the error sits). tools/aniso_probe.py.
            const float c = fma3(re.w.x, re.q.x, re.w.y, re.q.y, re.w.z, re.q.z) * re.rl;
            re.w = {__builtin_fmaf(-c, re.q.x, re.w.x), __builtin_fmaf(-c, re.q.y, re.w.y), __builtin_fmaf(-c, re.q.z, re.w.z)};
        }
#endif
#if LFS_REC_LOG2
        const float araw = re.vis;
#else
        const float vis = re.vis, opac = rec.r3.x;
        const float araw = opac * vis;
#endif
        const float alpha = fminf(0.999f, araw);
        LFS_EMUL_COUNT(2);
#if LFS_SEL_E64
        const lmask_t vmask = mask_le_i32_uniform(e.y, bin_final) & mask_nlt_f32(alpha, 1.f / 255.f); // (lane-mask form: lfs_raster_common.cuh)
        if (vmask == 0ull) return;
#else
        const bool valid = e.y <= bin_final && !(alpha < (1.f / 255.f)); // (inactive lanes carry bin_final = -1; vis > 1 cannot happen)
        if (__ballot(valid) == 0ull) return;
#endif
        LFS_EMUL_COUNT(3);
#if LFS_SEL_E64
        LFS_EMUL_LANES(vmask);
#else
        LFS_EMUL_LANES(__ballot(valid));
#endif

        // Invalid lanes are masked by zeroing three scalars (fac, v_op, and through it s): every reduced value below is
        // a product with one of them. (All factors are finite for an inactive lane: its direction is 0, so w = gro, t = 0.)
#if LFS_BWD_ALPHA0
        // an invalid lane takes part with alpha = 0: 1 / (1 - 0) = 1 exactly (v_rcp_f32 is exact at 1: tests/test_gpu_raster.py), so its T is multiplied by 1 and its
        // fac is 0 * T - one select instead of the two on T and fac
        const float alpha_v = valid ? alpha : 0.f;
        const float ra = fast_rcp(1.f - alpha_v);
        const float Tn = T * ra;
        T = Tn;
        const float fac = alpha_v * Tn;
#else
        const float ra = fast_rcp(1.f - alpha);
        const float Tn = T * ra;
#if LFS_SEL_E64
        T = sel_mask(vmask, Tn, T);
        const float fac = sel_mask(vmask, alpha * Tn, 0.f);
#else
        T = valid ? Tn : T;
        const float fac = valid ? alpha * Tn : 0.f;
#endif
#endif
        float v[16], v_extra = 0.f, cv;
        if (CDIM <= 3) {
            cv = rec.r3.y * vc[0];
            if (CDIM > 1) cv = __builtin_fmaf(rec.r3.z, vc[1], cv);
            if (CDIM > 2) cv = __builtin_fmaf(rec.r3.w, vc[2], cv);
        } else {
            const float* cp = colors + size_t(e.x) * CDIM;
            cv = cp[0] * vc[0];
#pragma unroll
            for (int k = 1; k < CDIM; ++k) cv = __builtin_fmaf(cp[k], vc[k], cv);
        }
        // dL/dalpha = (tail - B) / (1 - alpha) + T (c . v_c), B = sum over the entries behind of fac_j (c_j . v_c)
#if LFS_BWD_ALPHA0
        const float v_alpha = __builtin_fmaf(ra, tail, Tn * cv);   // `tail` carries tail - B: one fma per entry instead of a subtraction and an fma
        tail = __builtin_fmaf(-fac, cv, tail);
#else
        const float v_alpha = __builtin_fmaf(ra, tail - Bsum, Tn * cv);
        Bsum = __builtin_fmaf(fac, cv, Bsum);
#endif
#pragma unroll
        for (int k = 0; k < CDIM; ++k) {
            const float vrgb = fac * vc[k];
            if (k < 3) v[13 + k] = vrgb; else v_extra = vrgb;
        }
#pragma unroll
        for (int k = CDIM; k < 3; ++k) v[13 + k] = 0.f;
        // through alpha = min(0.999, opac * vis): no gradient on the clamped side
#if LFS_REC_LOG2
#if LFS_SEL_E64
        const float sgeo = sel_mask(vmask & mask_le_f32(araw, 0.999f), araw * v_alpha, 0.f);
#else
        const float sgeo = (valid && araw <= 0.999f) ? araw * v_alpha : 0.f; // s = alpha_raw * dL/dalpha = opac * dL/dopacity = vis * dL/dvis
#endif
        const float v_op = sgeo;                                            // (slot 12 holds opac * dL/dopacity: the finish kernels divide)
        v[12] = v_op;
#else
        const float v_op = (valid && araw <= 0.999f) ? vis * v_alpha : 0.f; // dL/dopacity
        v[12] = v_op;
        const float sgeo = opac * v_op;                                     // s = vis * dL/dvis
#endif
#if LFS_BWD_PK
        if (CDIM == 3 && MODE == RAY_GLOBAL) { // the 18 products and the first two reduction levels on register PAIRS (v_pk_mul_f32 / v_pk_add_f32)
            v2f V[8];
#if LFS_ACC_SYM
            // B'' = a (x) w (symmetric: six products) and a = s w - lfs_raster_common.cuh, LFS_ACC_SYM. Slots: xx, yy | xz, yz | xy, zz | ax, ay | az
            const v2f wxy = v2f{re.w.x, re.w.y};
            const v2f axy = wxy * sgeo;
            const float az = re.w.z * sgeo;
            V[0] = axy * wxy;
            V[1] = axy * re.w.z;
            V[2] = v2f{axy.x * wxy.y, az * re.w.z};
            V[3] = axy;
            V[4] = v2f{az, 0.f};
            V[5] = v2f{0.f, 0.f};
#elif LFS_REC_ROT
            // w = (-t q.x, -t q.y, G u): a = s w straight from q (one packed product for the two components that are multiples of q)
            // Slots 9 .. 11 of the row hold (a.z, a.x, a.y) in this form - the pair (a.x, a.y) is used as it comes out of the packed product; finish_geometry puts them back.
            const float nst = -(sgeo * re.t);
            const v2f axy = v2f{re.q.x, re.q.y} * nst;
            const float az = re.w.z * sgeo;
            const v2f vgxy = axy * re.t;
            const float vgz = az * re.t;
            V[0] = v2f{rd.x, rd.y} * vgxy.x;
            V[1] = v2f{vgxy.x * rd.z, vgxy.y * rd.x};
            V[2] = v2f{rd.y, rd.z} * vgxy.y;
            V[3] = v2f{rd.x, rd.y} * vgz;
            V[4] = v2f{vgz * rd.z, az};
            V[5] = axy;
#else
            const float ax = re.w.x * sgeo;
            const v2f ayz = v2f{re.w.y, re.w.z} * sgeo;
            const float vgx = ax * re.t;
            const v2f vgyz = ayz * re.t;
            V[0] = v2f{rd.x, rd.y} * vgx;
            V[1] = v2f{vgx * rd.z, vgyz.x * rd.x};
            V[2] = v2f{rd.y, rd.z} * vgyz.x;
            V[3] = v2f{rd.x, rd.y} * vgyz.y;
            V[4] = v2f{vgyz.y * rd.z, ax};
            V[5] = ayz;
#endif
            V[6] = v2f{v_op, fac * vc[0]};
            V[7] = v2f{vc[1], vc[2]} * fac;
#if LFS_BWD_LDS_REDUCE && LFS_RED_QUAD_ASM
#if LFS_RED_BUF_ATOMIC
            wave_sum16_atomic_quad<ACC, RED_SKIP>(V, acc + size_t(uint32_t(e.x)) * ACC_STRIDE, lane, red_base, red_rd, red_atomic_lane, ACC == 2 ? mse.det64 + size_t(e.x) * ACC_STRIDE : nullptr,
                                                  &red_buf, uint32_t(e.x) << 6);
#else
            wave_sum16_atomic_quad<ACC, RED_SKIP>(V, acc + size_t(uint32_t(e.x)) * ACC_STRIDE, lane, red_base, red_rd, red_atomic_lane, ACC == 2 ? mse.det64 + size_t(e.x) * ACC_STRIDE : nullptr);
#endif
#elif LFS_BWD_LDS_REDUCE
            wave_sum16_atomic_lds<ACC>(V, acc + size_t(e.x) * ACC_STRIDE, lane, red_scratch, ACC == 2 ? mse.det64 + size_t(e.x) * ACC_STRIDE : nullptr);
#else
            wave_sum16_atomic_pk<ACC>(V, acc + size_t(e.x) * ACC_STRIDE, lane, ACC == 2 ? mse.det64 + size_t(e.x) * ACC_STRIDE : nullptr);
#endif
            return;
        }
#endif
        const f3 a = re.w * sgeo;                                           // = -dL/dgro (the sign is undone in raster_finish_kernel)
        if (LFS_ACC_SYM && MODE == RAY_GLOBAL) { // the symmetric row (see the packed form above): B'' xx, yy, xz, yz, xy, zz | a | - - -
            v[0] = a.x * re.w.x; v[1] = a.y * re.w.y; v[2] = a.x * re.w.z; v[3] = a.y * re.w.z; v[4] = a.x * re.w.y; v[5] = a.z * re.w.z;
            v[6] = a.x; v[7] = a.y; v[8] = a.z; v[9] = 0.f; v[10] = 0.f; v[11] = 0.f;
            wave_sum16_atomic<ACC>(v, acc + size_t(e.x) * ACC_STRIDE, lane, ACC == 2 ? mse.det64 + size_t(e.x) * ACC_STRIDE : nullptr);
            if (CDIM > 3) {
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) v_extra += __shfl_xor(v_extra, m, 64);
                if (lane == 0) unsafeAtomicAdd(v_colors_extra + size_t(e.x) * CDIM + 3, v_extra);
            }
            return;
        }
        const f3 vg = a * re.t;                                             // dL/dq, q = (record matrix) d
        // dL/d(record matrix) = vg (x) d  [- a (x) (o - mu) per pixel only when the origin varies]
        v[0] = vg.x * rd.x; v[1] = vg.x * rd.y; v[2] = vg.x * rd.z;
        v[3] = vg.y * rd.x; v[4] = vg.y * rd.y; v[5] = vg.y * rd.z;
        v[6] = vg.z * rd.x; v[7] = vg.z * rd.y; v[8] = vg.z * rd.z;
        if (MODE == RAY_ROLLING) {
            const f3& om = re.om;
            v[0] -= a.x * om.x; v[1] -= a.x * om.y; v[2] -= a.x * om.z;
            v[3] -= a.y * om.x; v[4] -= a.y * om.y; v[5] -= a.y * om.z;
            v[6] -= a.z * om.x; v[7] -= a.z * om.y; v[8] -= a.z * om.z;
        }
        if (LFS_REC_ROT && MODE == RAY_GLOBAL) { v[9] = a.z; v[10] = a.x; v[11] = a.y; } // (the slot order of the rotated records: see the packed form above)
        else { v[9] = a.x; v[10] = a.y; v[11] = a.z; }
        wave_sum16_atomic<ACC>(v, acc + size_t(e.x) * ACC_STRIDE, lane, ACC == 2 ? mse.det64 + size_t(e.x) * ACC_STRIDE : nullptr);
        if (CDIM > 3) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v_extra += __shfl_xor(v_extra, m, 64);
            if (lane == 0) unsafeAtomicAdd(v_colors_extra + size_t(e.x) * CDIM + 3, v_extra);
        }
    };
    walk_cell_list<-1>(cl, recs, n_walk - 1, n_walk, eval, []() { return true; });
}

// dL/d(mean, quaternion, scale) of ONE Gaussian under ONE camera from its accumulator sums A = dL/d(record matrix), G = dL/dg - the vjp tail of Bwd.cu:318-333 through
// M = S^-1 R^T (global shutter: record matrix M Rinv, g = M (o - mu)). Shared by raster_finish_kernel, raster_finish_adam_kernel and gut_tail_kernel, which are held to
// each other bit for bit (tests/test_gpu_gut_step.py): the arithmetic is pinned contraction-free here, helpers included (lfs_math.cuh's take the including file's default and
// this file allows contraction - round 6: a third inlined copy of the fused form came out one ulp off in dL/dmeans, the compiler fuses a*b + c*d differently per context).
// ADDS to vm / vq / vs (the operator form sums over cameras).
template <bool UNIFORM_ORIGIN>
LFS_DI void finish_geometry(const float4 q, const float (&is)[3], const float (&A_in)[9], const f3 G_in, const f3 mu, const CamDev& cam, float (&vm)[3], float (&vq)[4], float (&vs)[3]) {
#pragma clang fp contract(off)
    m3 R;
    float qw = q.x, qx = q.y, qy = q.z, qz = q.w;
    const float qinv = 1.f / sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
    qx *= qinv; qy *= qinv; qz *= qinv; qw *= qinv;
    {   // quat_to_rotmat
        const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
        R.m[0][0] = 1.f - 2.f * (y2 + z2); R.m[1][0] = 2.f * (xy + wz); R.m[2][0] = 2.f * (xz - wy);
        R.m[0][1] = 2.f * (xy - wz); R.m[1][1] = 1.f - 2.f * (x2 + z2); R.m[2][1] = 2.f * (yz + wx);
        R.m[0][2] = 2.f * (xz + wy); R.m[1][2] = 2.f * (yz - wx); R.m[2][2] = 1.f - 2.f * (x2 + y2);
    }
    m3 M;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M.m[r][c] = is[r] * R.m[c][r];
#if LFS_ACC_SYM
    if (UNIFORM_ORIGIN) { // the symmetric row (lfs_raster_common.cuh, LFS_ACC_SYM): A_in = B'' (xx, yy, xz, yz, xy, zz) | sum a'', every slot times REC_UNSCALE by the caller; G_in: empty slots
        const float omx = cam.origin.x - mu.x, omy = cam.origin.y - mu.y, omz = cam.origin.z - mu.z;
        const f3 gv{M.m[0][0] * omx + M.m[0][1] * omy + M.m[0][2] * omz, M.m[1][0] * omx + M.m[1][1] * omy + M.m[1][2] * omz, M.m[2][0] * omx + M.m[2][1] * omy + M.m[2][2] * omz};
        m3 U;
        rot_frame_f32(gv, U);
        float Bp[3][3];   // B'' = sum a'' (x) w'', two factors of c in it: the second REC_UNSCALE here
        Bp[0][0] = A_in[0] * REC_UNSCALE; Bp[1][1] = A_in[1] * REC_UNSCALE; Bp[2][2] = A_in[5] * REC_UNSCALE;
        Bp[0][2] = Bp[2][0] = A_in[2] * REC_UNSCALE; Bp[1][2] = Bp[2][1] = A_in[3] * REC_UNSCALE; Bp[0][1] = Bp[1][0] = A_in[4] * REC_UNSCALE;
        float T[3][3], B[3][3];   // B = U^T B'' U: the sums in the Gaussian's own frame
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) T[k][c] = Bp[k][0] * U.m[0][c] + Bp[k][1] * U.m[1][c] + Bp[k][2] * U.m[2][c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) B[r][c] = U.m[0][r] * T[0][c] + U.m[1][r] * T[1][c] + U.m[2][r] * T[2][c];
        const float as[3] = {U.m[0][0] * A_in[6] + U.m[1][0] * A_in[7] + U.m[2][0] * A_in[8], U.m[0][1] * A_in[6] + U.m[1][1] * A_in[7] + U.m[2][1] * A_in[8],
                             U.m[0][2] * A_in[6] + U.m[1][2] * A_in[7] + U.m[2][2] * A_in[8]};   // sum a = -dL/dg
        // mean: g = M (o - mu)  ->  dL/dmu = -M^T dL/dg = M^T sum a
        vm[0] += M.m[0][0] * as[0] + M.m[1][0] * as[1] + M.m[2][0] * as[2];
        vm[1] += M.m[0][1] * as[0] + M.m[1][1] * as[1] + M.m[2][1] * as[2];
        vm[2] += M.m[0][2] * as[0] + M.m[1][2] * as[1] + M.m[2][2] * as[2];
        // dL/dM = -B M^-T = -B S R^T. Scales: dL/ds_c = -(1 / s_c^2) (dL/dM R)_cc = B_cc / s_c - a sum of like-signed terms, where dL/dA . M + dL/dg . g cancelled to 1e-6.
        // Rotation: dL/dR[r][c] = dL/dM[c][r] / s_c = -(1 / s_c) sum_k B[c][k] s_k R[r][k]
        const float sk[3] = {1.f / is[0], 1.f / is[1], 1.f / is[2]};
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) g[r][c] = -is[c] * (B[c][0] * sk[0] * R.m[r][0] + B[c][1] * sk[1] * R.m[r][1] + B[c][2] * sk[2] * R.m[r][2]);
        {   // quat_to_rotmat_vjp on the normalised quaternion
            const float vw = 2.f * (qx * (g[2][1] - g[1][2]) + qy * (g[0][2] - g[2][0]) + qz * (g[1][0] - g[0][1]));
            const float vx = 2.f * (-2.f * qx * (g[1][1] + g[2][2]) + qy * (g[1][0] + g[0][1]) + qz * (g[2][0] + g[0][2]) + qw * (g[2][1] - g[1][2]));
            const float vy = 2.f * (qx * (g[1][0] + g[0][1]) - 2.f * qy * (g[0][0] + g[2][2]) + qz * (g[2][1] + g[1][2]) + qw * (g[0][2] - g[2][0]));
            const float vz = 2.f * (qx * (g[2][0] + g[0][2]) + qy * (g[2][1] + g[1][2]) - 2.f * qz * (g[0][0] + g[1][1]) + qw * (g[1][0] - g[0][1]));
            const float dq = vw * qw + vx * qx + vy * qy + vz * qz;
            vq[0] += (vw - dq * qw) * qinv; vq[1] += (vx - dq * qx) * qinv; vq[2] += (vy - dq * qy) * qinv; vq[3] += (vz - dq * qz) * qinv;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) vs[c] += B[c][c] * is[c];
        return;
    }
    const float (&A)[9] = A_in; const f3 G = G_in;
#elif LFS_REC_ROT
    float Au[9]; f3 Gu = G_in;
#pragma unroll
    for (int k = 0; k < 9; ++k) Au[k] = A_in[k];
    if (UNIFORM_ORIGIN) { // the sums were accumulated in the record's rotated frame (lfs_raster_common.cuh, LFS_REC_ROT): dL/dA = U^T dL/dA'', dL/dg = U^T dL/dg''
        const float omx = cam.origin.x - mu.x, omy = cam.origin.y - mu.y, omz = cam.origin.z - mu.z;
        const f3 g{M.m[0][0] * omx + M.m[0][1] * omy + M.m[0][2] * omz, M.m[1][0] * omx + M.m[1][1] * omy + M.m[1][2] * omz, M.m[2][0] * omx + M.m[2][1] * omy + M.m[2][2] * omz};
        RotFrame F;
        rot_frame(g, F);   // (the record's frame, bit for bit: lfs_raster_common.cuh)
#pragma unroll
        for (int c = 0; c < 3; ++c) rot_apply_t(F, A_in[c], A_in[3 + c], A_in[6 + c], Au[c], Au[3 + c], Au[6 + c]);
        rot_apply_t(F, G_in.y, G_in.z, G_in.x, Gu.x, Gu.y, Gu.z);   // slots 9 .. 11 arrive as (z, x, y) (raster_bwd_kernel)
    }
    const float (&A)[9] = Au; const f3 G = Gu;
#else
    const float (&A)[9] = A_in; const f3 G = G_in;
#endif
    // dL/dM (math rows r, cols c). Global shutter: the record matrix was M Rinv, so dL/dM = dL/d(M Rinv) Rinv^T
    m3 vM;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (UNIFORM_ORIGIN) {
                const m3& Ri = cam.Rinv;
                vM.m[r][c] = A[3 * r] * Ri.m[c][0] + A[3 * r + 1] * Ri.m[c][1] + A[3 * r + 2] * Ri.m[c][2];
            } else vM.m[r][c] = A[3 * r + c];
        }
    if (UNIFORM_ORIGIN) {
        const float gv[3] = {G.x, G.y, G.z}, ov[3] = {cam.origin.x - mu.x, cam.origin.y - mu.y, cam.origin.z - mu.z};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) vM.m[r][c] += gv[r] * ov[c];
    }
    // mean: gro = M (o - mu)  ->  dL/dmu = -M^T G
    vm[0] -= M.m[0][0] * G.x + M.m[1][0] * G.y + M.m[2][0] * G.z;
    vm[1] -= M.m[0][1] * G.x + M.m[1][1] * G.y + M.m[2][1] * G.z;
    vm[2] -= M.m[0][2] * G.x + M.m[1][2] * G.y + M.m[2][2] * G.z;
    // M = S^-1 R^T, i.e. P = M^T = R S^-1 with dL/dP = (dL/dM)^T:  dL/dR[r][c] = dL/dP[r][c] / s_c ;  dL/ds_c = -(1/s_c^2) sum_r R[r][c] dL/dP[r][c]
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) g[r][c] = vM.m[c][r] * is[c];
    {   // quat_to_rotmat_vjp on the normalised quaternion
        const float vw = 2.f * (qx * (g[2][1] - g[1][2]) + qy * (g[0][2] - g[2][0]) + qz * (g[1][0] - g[0][1]));
        const float vx = 2.f * (-2.f * qx * (g[1][1] + g[2][2]) + qy * (g[1][0] + g[0][1]) + qz * (g[2][0] + g[0][2]) + qw * (g[2][1] - g[1][2]));
        const float vy = 2.f * (qx * (g[1][0] + g[0][1]) - 2.f * qy * (g[0][0] + g[2][2]) + qz * (g[2][1] + g[1][2]) + qw * (g[0][2] - g[2][0]));
        const float vz = 2.f * (qx * (g[2][0] + g[0][2]) + qy * (g[2][1] + g[1][2]) - 2.f * qz * (g[0][0] + g[1][1]) + qw * (g[1][0] - g[0][1]));
        const float dq = vw * qw + vx * qx + vy * qy + vz * qz;
        vq[0] += (vw - dq * qw) * qinv; vq[1] += (vx - dq * qx) * qinv; vq[2] += (vy - dq * qy) * qinv; vq[3] += (vz - dq * qz) * qinv;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        vs[c] += -is[c] * is[c] * (R.m[0][c] * vM.m[c][0] + R.m[1][c] * vM.m[c][1] + R.m[2][c] * vM.m[c][2]);
}

// ---------------------------------------------------------------------------
// finish: accumulator -> dL/d(means, quats, scales, colors, opacities)
// ---------------------------------------------------------------------------
template <bool UNIFORM_ORIGIN>
__global__ void __launch_bounds__(256) raster_finish_kernel(
    const uint32_t C, const uint32_t N, const uint32_t channels,
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const CamDev* __restrict__ cams, const float* __restrict__ acc,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_colors, float* __restrict__ v_opacities, const float* __restrict__ loss_slots, float* __restrict__ loss,
    const float* __restrict__ opacities) {
    if (loss_slots != nullptr && blockIdx.x == 0) { // fused MSE: fold the backward kernel's partial sums into the caller's accumulator
        float v = threadIdx.x < LOSS_SLOTS ? loss_slots[threadIdx.x] : 0.f;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0 && v != 0.f) unsafeAtomicAdd(loss, v);
    }
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N) return;
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    bool geom_loaded = false;
    f3 mu{0.f, 0.f, 0.f}; float4 q = make_float4(1.f, 0.f, 0.f, 0.f); float is[3] = {0.f, 0.f, 0.f};
    for (uint32_t cid = 0; cid < C; ++cid) {
        const size_t idx = size_t(cid) * N + gid;
        const float4* a4 = reinterpret_cast<const float4*>(acc + idx * ACC_STRIDE);
        float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3];
#if LFS_REC_LOG2
        { // the row was accumulated against the scaled record (lfs_raster_common.cuh): A' = c A, G' = c G, slot 12 = opac * dL/dopac
            a0.x *= REC_UNSCALE; a0.y *= REC_UNSCALE; a0.z *= REC_UNSCALE; a0.w *= REC_UNSCALE; a1.x *= REC_UNSCALE; a1.y *= REC_UNSCALE; a1.z *= REC_UNSCALE;
            a1.w *= REC_UNSCALE; a2.x *= REC_UNSCALE; a2.y *= REC_UNSCALE; a2.z *= REC_UNSCALE; a2.w *= REC_UNSCALE;
            const float op = opacities[idx];
            a3.x = a3.x != 0.f ? a3.x / op : 0.f;
        }
#endif
        v_opacities[idx] = a3.x;
        float* vcol = v_colors + idx * channels;
        vcol[0] = a3.y;
        if (channels > 1) vcol[1] = a3.z;
        if (channels > 2) vcol[2] = a3.w;
        // (channel 3, when present, was accumulated straight into v_colors by the bwd kernel)
        const float A[9] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x};
        const f3 G{-a2.y, -a2.z, -a2.w}; // the bwd kernel accumulates -dL/dgro
        bool any = G.x != 0.f || G.y != 0.f || G.z != 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) any |= A[k] != 0.f;
        if (!any) continue;
        if (!geom_loaded) {
            mu = {means[3 * gid], means[3 * gid + 1], means[3 * gid + 2]};
            q = reinterpret_cast<const float4*>(quats)[gid];
            is[0] = 1.f / scales[3 * gid]; is[1] = 1.f / scales[3 * gid + 1]; is[2] = 1.f / scales[3 * gid + 2];
            geom_loaded = true;
        }
        finish_geometry<UNIFORM_ORIGIN>(q, is, A, G, mu, cams[cid], vm, vq, vs);
    }
    v_means[3 * gid] = vm[0]; v_means[3 * gid + 1] = vm[1]; v_means[3 * gid + 2] = vm[2];
    reinterpret_cast<float4*>(v_quats)[gid] = make_float4(vq[0], vq[1], vq[2], vq[3]);
    v_scales[3 * gid] = vs[0]; v_scales[3 * gid + 1] = vs[1]; v_scales[3 * gid + 2] = vs[2];
}

// The all-inline training step (ONE camera, global shutter, one view per step on one rank): raster_finish_kernel + the activation backward
// (l2_fused.hip: normalize / exp / sigmoid vjp + the MCMC regularisers) + the Adam updates of means, raw scales, raw quaternions and raw
// opacities (adam.hip) in ONE pass over the Gaussians - no gradient tensor is written or re-read. acc row: dL/dA (9) | -dL/dg (3) | dL/dopacity |
// dL/dcolour (3: consumed by lfs_sh_model_bwd_adam_all, which hands back dL/d(dirs) in v_dirs). Same operations in the same order as
// the three separate kernels (the activation and Adam arithmetic is un-fused there: pinned with fp contract(off) here).
struct FinishAdam { float* m[4]; float* v[4]; AdamScalars s[4]; float scale_reg, opacity_reg; }; // order: means, raw_scales, raw_quats, raw_opacities
// ADAM = false (multi-GPU / multi-view / non-MSE steps, which need gradient TENSORS): the same pass up to the raw-parameter gradients, written (or added,
// `accumulate`) to g_* instead of being consumed - raster_finish_kernel + activations_bwd_kernel + the copy of dL/dmeans in one launch; dL/dcolour goes
// to v_colors for the SH backward, which adds dL/d(dirs) onto g_means afterwards. *loss += the fused MSE (as raster_finish_kernel).
struct FinishGrads { float* g_means; float* g_scales; float* g_quats; float* g_opac; float* v_colors; int accumulate; };
#ifndef LFS_FINISH_ONE_TRIP
#define LFS_FINISH_ONE_TRIP 1   // (round 6) every load of raster_finish_adam_kernel in one round trip; 0 = the round-5 form (four dependent trips), kept for the A/B
#endif
#if LFS_FINISH_ONE_TRIP && (!LFS_REC_LOG2 || !LFS_FINISH_LDS_ROWS)
#error "LFS_FINISH_ONE_TRIP is written for the LFS_REC_LOG2 records and the LDS row hand-over"
#endif
#ifndef LFS_FINISH_BLOCK
#define LFS_FINISH_BLOCK 256   // threads per workgroup of raster_finish_adam_kernel (A/B hook: 512 / 1024 = a larger contiguous chunk per stream and CU for the 29-stream pass)
#endif
// the fused-loss fold reads LOSS_SLOTS = 256 partial sums with the first four wavefronts of workgroup 0 and adds wave_sum[0..3]: smaller workgroups would drop slots
static_assert(LFS_FINISH_BLOCK >= 256 && LFS_FINISH_BLOCK % 64 == 0 && LFS_FINISH_BLOCK <= 1024, "LFS_FINISH_BLOCK: 256, 320, ..., 1024");
template <bool ADAM>
__global__ void __launch_bounds__(LFS_FINISH_BLOCK) raster_finish_adam_kernel(
    const uint32_t N, float* __restrict__ means, float* __restrict__ raw_scales, float* __restrict__ raw_quats, float* __restrict__ raw_opacities,
    const float* __restrict__ quats, const float* __restrict__ scales, const float* __restrict__ opacities,
    const CamDev* __restrict__ cams, const float* __restrict__ acc, const float* __restrict__ v_dirs, const FinishAdam ad, const FinishGrads gr,
    const float* __restrict__ loss_slots, float* __restrict__ loss, const int32_t* __restrict__ abort_flag = nullptr) {
#if LFS_FINISH_ONE_TRIP && !defined(LFS_EMULATE)   // (the emulator has no wave-private LDS hand-over: it runs the round-5 form below)
    // Round 6: ONE memory round trip per wavefront. The round-5 form had four in series - abort-flag pointer -> flag -> the accumulator rows (their LDS hand-over
    // sits behind a scheduling barrier no load may cross) -> the other 21 loads (two of them issued late, behind the first arithmetic) - in a kernel whose 29 streams
    // reach 5.1 TB/s in a trivial pass (tools/hbm_stream.hip: multi_rmw_32_streams) and that ran at 3.1 - 3.7. Here every load of the pass is issued before the first
    // wait, through (SGPR base, 32-bit byte offset) addresses - three offset registers instead of a 64-bit address pair per stream - and the abort flag is looked at
    // where it matters: in front of the stores.
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = gid < N;
    const uint32_t g = live ? gid : (N - 1u);              // (tail lanes read the last row: no branch around the loads)
    const uint32_t o4 = g << 2, o12 = g * 12u, o16 = g << 4;
    auto at = [](const void* base, uint32_t byte_off) { return reinterpret_cast<const char*>(base) + byte_off; };
    auto ld3o = [&](const float* base, float (&dst)[3]) { const V3f t = *reinterpret_cast<const V3f*>(at(base, o12)); dst[0] = t.a[0]; dst[1] = t.a[1]; dst[2] = t.a[2]; };
    auto ld4o = [&](const float* base) { return *reinterpret_cast<const float4*>(at(base, o16)); };
    auto ld1o = [&](const float* base) { return *reinterpret_cast<const float*>(at(base, o4)); };
    int32_t aborted = 0;
    if (ADAM && abort_flag != nullptr) aborted = *abort_flag; // (scalar load: in flight with everything else)
    // the 64 accumulator rows of a wavefront (4 KB contiguous) as four fully coalesced 1-KB loads, handed to their lanes through a wave-private LDS block below
    __shared__ float4 s_rows[LFS_FINISH_BLOCK / 64][64 * 5];
    float4* const rows = s_rows[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63, g0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane;
    float4 t_rows[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { // (rows past the end: the last row again - an address clamp instead of a branch around the load)
        const uint32_t idx = i * 64 + lane, row = min(g0 + (idx >> 2), N - 1u);
        t_rows[i] = *reinterpret_cast<const float4*>(at(acc, (row << 6) + ((idx & 3u) << 4)));
    }
    float mu_a[3], sc[3], vd[3] = {0.f, 0.f, 0.f};
    ld3o(means, mu_a); ld3o(scales, sc);
    if (ADAM || v_dirs != nullptr) ld3o(v_dirs, vd);   // (gradient-tensor form: nullable - the SH backward then adds dL/d(dirs) onto g_means afterwards)
    const float4 q = ld4o(quats), rq = ld4o(raw_quats);
    const float o = ld1o(opacities);
    float m0[3] = {0.f, 0.f, 0.f}, v0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f}, m1[3] = {0.f, 0.f, 0.f}, v1[3] = {0.f, 0.f, 0.f};
    float4 mq = make_float4(0.f, 0.f, 0.f, 0.f), vq4 = mq;
    float po = 0.f, mo = 0.f, vo = 0.f;
    if (ADAM) {
        ld3o(ad.m[0], m0); ld3o(ad.v[0], v0); ld3o(raw_scales, p1); ld3o(ad.m[1], m1); ld3o(ad.v[1], v1);
        mq = ld4o(ad.m[2]); vq4 = ld4o(ad.v[2]);
        po = ld1o(raw_opacities); mo = ld1o(ad.m[3]); vo = ld1o(ad.v[3]);
    }
    float loss_v = 0.f;
    if (loss_slots != nullptr && blockIdx.x == 0) loss_v = threadIdx.x < LOSS_SLOTS ? loss_slots[threadIdx.x] : 0.f;   // (LOSS_SLOTS = 256: the first four wavefronts carry the slots)
    // ---- everything is in flight; from here on the pass only consumes (the accumulator rows were requested first and are the first to be waited for) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint32_t idx = i * 64 + lane; rows[(idx >> 2) * 5 + (idx & 3)] = t_rows[i]; }
    __builtin_amdgcn_wave_barrier();
    if (loss_slots != nullptr && blockIdx.x == 0) { // *loss = the fused MSE (a store in a fixed order: the step needs no zeroed accumulator)
        __shared__ float wave_sum[LFS_FINISH_BLOCK / 64];
        float v = loss_v;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0 && aborted == 0) {
            const float total = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
            if (ADAM) *loss = total; else if (total != 0.f) unsafeAtomicAdd(loss, total);
        }
    }
    if (!live || aborted != 0) return; // (aborted: uniform) speculative step that did not fit its buffers: no update, the host runs it again
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    const float4* a4 = rows + lane * 5;
    const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3];
    const float A[9] = {a0.x * REC_UNSCALE, a0.y * REC_UNSCALE, a0.z * REC_UNSCALE, a0.w * REC_UNSCALE, a1.x * REC_UNSCALE, a1.y * REC_UNSCALE, a1.z * REC_UNSCALE,
                        a1.w * REC_UNSCALE, a2.x * REC_UNSCALE};
    const f3 G{-a2.y * REC_UNSCALE, -a2.z * REC_UNSCALE, -a2.w * REC_UNSCALE};
    bool any = G.x != 0.f || G.y != 0.f || G.z != 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) any |= A[k] != 0.f;
    const f3 mu{mu_a[0], mu_a[1], mu_a[2]};
    auto st3a = [&](float* base, const float (&src)[3]) { V3f t; t.a[0] = src[0]; t.a[1] = src[1]; t.a[2] = src[2]; *reinterpret_cast<V3f*>(const_cast<char*>(at(base, o12))) = t; };
    auto ld3a = [&](const float* base, float (&dst)[3]) { ld3o(base, dst); };
    const float v_opac = a3.x != 0.f ? a3.x / o : 0.f;
#else
    if (ADAM && abort_flag != nullptr && *abort_flag != 0) return; // (uniform) speculative step that did not fit its buffers: no update, the host runs it again
    if (loss_slots != nullptr && blockIdx.x == 0) { // *loss = the fused MSE (a store in a fixed order: the step needs no zeroed accumulator)
        __shared__ float wave_sum[LFS_FINISH_BLOCK / 64];
        float v = threadIdx.x < LOSS_SLOTS ? loss_slots[threadIdx.x] : 0.f;   // (LOSS_SLOTS = 256: the first four wavefronts carry the slots whatever the block size)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float total = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
            if (ADAM) *loss = total; else if (total != 0.f) unsafeAtomicAdd(loss, total);
        }
    }
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
#if LFS_FINISH_LDS_ROWS && !defined(LFS_EMULATE)   // (the emulator switches lanes only at cross-lane operations: no wave-private LDS hand-over there)
    // the 64 accumulator rows of a wavefront (4 KB contiguous) as four fully coalesced 1-KB loads, handed to their lanes through a wave-private LDS block
    // (row stride 20 floats: 16-byte aligned, conflict-free on the read side) instead of four 16-byte loads per lane at a 64-byte stride
    __shared__ float4 s_rows[LFS_FINISH_BLOCK / 64][64 * 5];
    float4* const rows = s_rows[threadIdx.x >> 6];
    {
        const uint32_t lane = threadIdx.x & 63, g0 = gid - lane;
        const float4* src = reinterpret_cast<const float4*>(acc + size_t(g0) * ACC_STRIDE);
        float4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t idx = i * 64 + lane; t[i] = (g0 + (idx >> 2) < N) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t idx = i * 64 + lane; rows[(idx >> 2) * 5 + (idx & 3)] = t[i]; }
        __builtin_amdgcn_wave_barrier();
    }
    if (gid >= N) return;
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    const float4* a4 = rows + (threadIdx.x & 63) * 5;
    const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3];
#else
    if (gid >= N) return;
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    const float4* a4 = reinterpret_cast<const float4*>(acc + size_t(gid) * ACC_STRIDE);
    const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2], a3 = a4[3];
#endif
#if LFS_REC_LOG2
    // the row was accumulated against the scaled record (lfs_raster_common.cuh): A' = c A, G' = c G, slot 12 = opac * dL/dopac (divided where `o` is loaded)
    const float A[9] = {a0.x * REC_UNSCALE, a0.y * REC_UNSCALE, a0.z * REC_UNSCALE, a0.w * REC_UNSCALE, a1.x * REC_UNSCALE, a1.y * REC_UNSCALE, a1.z * REC_UNSCALE,
                        a1.w * REC_UNSCALE, a2.x * REC_UNSCALE};
    const f3 G{-a2.y * REC_UNSCALE, -a2.z * REC_UNSCALE, -a2.w * REC_UNSCALE};
#else
    const float A[9] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x};
    const f3 G{-a2.y, -a2.z, -a2.w};
#endif
    bool any = G.x != 0.f || G.y != 0.f || G.z != 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) any |= A[k] != 0.f;
    // (lfs_math.cuh ld3 / st3: three floats as one 12-byte access)
    auto ld3a = [&](const float* base, float (&dst)[3]) { const f3 t = ld3(base, gid); dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; };
    auto st3a = [&](float* base, const float (&src)[3]) { st3(base, gid, f3{src[0], src[1], src[2]}); };
    float sc[3], vd[3] = {0.f, 0.f, 0.f};
    const f3 mu = ld3(means, gid);
    ld3a(scales, sc);
    if (ADAM || v_dirs != nullptr) ld3a(v_dirs, vd);   // (gradient-tensor form: nullable - the SH backward then adds dL/d(dirs) onto g_means afterwards)
    // Every operand of the pass is requested HERE, before any arithmetic (round 3): the kernel used to fetch in three dependent phases (rows -> quaternion
    // inside `if (any)` -> moments), which left each wavefront with 3 - 7 loads in flight and the kernel at 3.5 TB/s. The geometry vjp below now runs for
    // every Gaussian and is SELECTED by `any` (un-touched Gaussians keep exact zeros; their 1/scale may be inf), so no load hides behind a branch.
    const float4 q = reinterpret_cast<const float4*>(quats)[gid];
    const float4 rq = reinterpret_cast<const float4*>(raw_quats)[gid];
    const float o = opacities[gid];
    const float v_opac = LFS_REC_LOG2 ? (a3.x != 0.f ? a3.x / o : 0.f) : a3.x;
    float m0[3] = {0.f, 0.f, 0.f}, v0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f}, m1[3] = {0.f, 0.f, 0.f}, v1[3] = {0.f, 0.f, 0.f};
    float4 mq = make_float4(0.f, 0.f, 0.f, 0.f), vq4 = mq;
    float po = 0.f, mo = 0.f, vo = 0.f;
    if (ADAM) {
        ld3a(ad.m[0], m0); ld3a(ad.v[0], v0); ld3a(raw_scales, p1); ld3a(ad.m[1], m1); ld3a(ad.v[1], v1);
        mq = reinterpret_cast<const float4*>(ad.m[2])[gid]; vq4 = reinterpret_cast<const float4*>(ad.v[2])[gid];
        po = raw_opacities[gid]; mo = ad.m[3][gid]; vo = ad.v[3][gid];
    }
#endif // LFS_FINISH_ONE_TRIP
    { // exactly raster_finish_kernel<true> for C == 1 (selected by `any` at the end)
        const float is[3] = {1.f / sc[0], 1.f / sc[1], 1.f / sc[2]};
        finish_geometry<true>(q, is, A, G, mu, cams[0], vm, vq, vs);
#pragma unroll
        for (int k = 0; k < 3; ++k) { vm[k] = any ? vm[k] : 0.f; vs[k] = any ? vs[k] : 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) vq[k] = any ? vq[k] : 0.f;
    }
    {
#pragma clang fp contract(off)
        // + dL/d(dirs) of the SH backward (sh.hip adds it onto the rasterizer's dL/dmeans in the separate path)
        float gm[3] = {vm[0] + vd[0], vm[1] + vd[1], vm[2] + vd[2]};
        // activations_bwd_kernel<false> (l2_fused.hip)
        const float nrm = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
        float gq[4];
        if (nrm > 1e-12f) {
            const float inv = 1.f / nrm;
            const float y[4] = {rq.x * inv, rq.y * inv, rq.z * inv, rq.w * inv};
            const float d = vq[0] * y[0] + vq[1] * y[1] + vq[2] * y[2] + vq[3] * y[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) gq[k] = (vq[k] - d * y[k]) * inv;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) gq[k] = vq[k] * 1e12f;
        }
        float gs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) gs[k] = (vs[k] + ad.scale_reg) * sc[k];
        const float go = (v_opac + ad.opacity_reg) * o * (1.f - o);
        if (!ADAM) { // gradient tensors out (activations_bwd_kernel's stores; dL/dmeans = the rasterizer's part, the SH backward adds the rest)
            const float v_col[3] = {a3.y, a3.z, a3.w};
            if (gr.v_colors != nullptr) st3a(gr.v_colors, v_col);
            float4* gqo = reinterpret_cast<float4*>(gr.g_quats) + gid;
            if (gr.accumulate) {
                float om[3], os[3];
                ld3a(gr.g_means, om); ld3a(gr.g_scales, os);
                const float4 oq = *gqo;
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2) { gm[k2] = om[k2] + gm[k2]; gs[k2] = os[k2] + gs[k2]; }
                gq[0] = oq.x + gq[0]; gq[1] = oq.y + gq[1]; gq[2] = oq.z + gq[2]; gq[3] = oq.w + gq[3];
                st3a(gr.g_means, gm); st3a(gr.g_scales, gs);
                *gqo = make_float4(gq[0], gq[1], gq[2], gq[3]);
                gr.g_opac[gid] += go;
            } else {
                st3a(gr.g_means, gm); st3a(gr.g_scales, gs);
                *gqo = make_float4(gq[0], gq[1], gq[2], gq[3]);
                gr.g_opac[gid] = go;
            }
            return;
        }
        // Adam (adam_multi_kernel's per-element update). Every moment is loaded before the first store (the moment arrays hang off a struct: no
        // __restrict__, a load could not move above an earlier store) and three-float rows move as 12-byte accesses (lfs_math.cuh). Measured on
        // one box against the element-by-element version: no difference (0.094 - 0.104 ms either way; the kernel walks 29 streams)
        float p[3] = {mu.x, mu.y, mu.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            adam_elem(p[k], m0[k], v0[k], gm[k], ad.s[0]);
            adam_elem(p1[k], m1[k], v1[k], gs[k], ad.s[1]);
        }
        float pq[4] = {rq.x, rq.y, rq.z, rq.w};
        adam_elem(pq[0], mq.x, vq4.x, gq[0], ad.s[2]); adam_elem(pq[1], mq.y, vq4.y, gq[1], ad.s[2]);
        adam_elem(pq[2], mq.z, vq4.z, gq[2], ad.s[2]); adam_elem(pq[3], mq.w, vq4.w, gq[3], ad.s[2]);
        adam_elem(po, mo, vo, go, ad.s[3]);
        st3a(means, p); st3a(ad.m[0], m0); st3a(ad.v[0], v0);
        st3a(raw_scales, p1); st3a(ad.m[1], m1); st3a(ad.v[1], v1);
        reinterpret_cast<float4*>(raw_quats)[gid] = make_float4(pq[0], pq[1], pq[2], pq[3]);
        reinterpret_cast<float4*>(ad.m[2])[gid] = mq; reinterpret_cast<float4*>(ad.v[2])[gid] = vq4;
        raw_opacities[gid] = po; ad.m[3][gid] = mo; ad.v[3][gid] = vo;
    }
}

// ---------------------------------------------------------------------------
// The TAIL of the all-inline training step in ONE pass over the Gaussians (round 6): SH backward + Adam(sh0, shN) + finish + activation backward +
// Adam(means, scales, quaternions, opacities) + - when the caller names the NEXT view - the SH colours of the next step. Replaces three launches of the one-call step
// (sh_bwd_kernel<.., ADAM>, raster_finish_adam_kernel<true>, next step's sh_fwd_kernel):
//   * dL/d(dirs) goes from the SH backward to the means update in registers (was: 12 B written + read per Gaussian);
//   * the next view's colours are evaluated from the coefficient rows while they sit in registers for their Adam update (was: sh_fwd re-reads 192 B per Gaussian, 65 us);
//   * the pass that follows the 1.1 GB read-modify-write of the SH Adam update no longer exists (finish_adam ran at 3.2 - 3.5 TB/s behind it, 5.9 alone: the dirty lines
//     of the update were still draining - profiles/r06/stream_kernels_alone_1M.json).
// One wavefront per 64 Gaussians, five phases (__syncthreads between them; LDS: basis rows, a second row block, the masked colour gradient):
//   1  lane = Gaussian          : direction of THIS view, visibility, basis -> lds_b; dL/dcolour (accumulator slots 13..15) under the clamp mask -> ldv; the finish pass's
//                                 operands are requested here and land under phase 2
//   2  lane = (Gaussian, basis) : s_k = coefficient row . dL/dcolour -> lds_s (rows of Gaussians without a colour gradient are not fetched)
//   3  lane = Gaussian          : dL/d(dirs) = sum_k s_k grad b_k; then exactly raster_finish_adam_kernel<true>'s arithmetic with it; the updated mean stays in registers
//   4  lane = Gaussian          : direction of the NEXT view from the updated mean, its basis -> lds_s
//   5  lane = (Gaussian, basis) : coefficient row + moments -> Adam -> stored; next colour = sum_k (new basis)_k x (updated row) -> colors (clamped as sh_fwd_kernel)
// The SH arithmetic is pinned contraction-free (lfs_sh.cuh and the blocks below) as in sh.hip; the finish arithmetic is raster_finish_adam_kernel's, statement by statement:
// parameters and moments come out bit-identical to the three kernels (tests/test_gpu_gut_step.py, tests/test_emulated_step_pack.py).
struct GutTail {
    uint32_t N, K; int degree;
    float *means, *sh0, *shN, *raw_scales, *raw_quats, *raw_opacities;
    const float *quats, *scales, *opacities;       // activated values (the projection kernel wrote them)
    const float* viewmat; const float* next_viewmat;   // next_viewmat: NULL = no colours for the next step
    const int32_t* radii; float* colors;             // in: this step's colours (clamp mask); out (next_viewmat given): the next step's, for EVERY Gaussian
    const float* acc;
    float *m0, *v0, *mN, *vN; AdamScalars s0, sN;    // sh0 / shN moments
    FinishAdam fin;
    const float* loss_slots; float* loss; const int32_t* abort_flag;
};
#ifndef LFS_TAIL_DEPTH
#define LFS_TAIL_DEPTH 4   // coefficient rows (parameter + two moments) in flight per lane in phase 5
#endif
// Same-box A/B with the order rotated, 4 rounds of 200 steps, SYN-B (profiles/r06/lease16_count_scan_tail_ab.txt; lease 15 before it: the same ranking on another box):
//   KEEP EARLY DEPTH   img/s (median)            KEEP EARLY DEPTH   img/s
//    0     0     4      670.3                     1     1     4      687.3   <- default
//    0     1     4      681.7                     1     1     8      687.9
//    0     0     8      686.2                     1     1     4 NT   686.8
//    0     1     8      682.7                     1     0     4      lease 15: below 0 / 0
#ifndef LFS_TAIL_KEEP
#define LFS_TAIL_KEEP 1    // 1: the coefficient rows phase 2 fetches stay in registers for phase 5 (same lane, same rows: 3 x LPG VGPRs; every row is then fetched in phase 2;
#endif                     //    196 - 224 VGPRs: two wavefronts per SIMD, each with 16 rows x 12 B per lane in flight in phase 2)
#ifndef LFS_TAIL_EARLY
#define LFS_TAIL_EARLY 1   // 1: the first LFS_TAIL_DEPTH moment rows of phase 5 are requested in front of phase 3 (they land under the finish arithmetic)
#endif
#ifndef LFS_TAIL_NT
#define LFS_TAIL_NT 0      // 1: the moments leave with non-temporal stores (nothing reads them for a whole step) - no difference measured
#endif
template <int LPG, bool NEXT>
__global__ void __launch_bounds__(64) gut_tail_kernel(const GutTail t, const CamDev* __restrict__ cams) {
    __shared__ float lds_b[64 * (LPG + 1)];
    __shared__ float lds_s[64 * (LPG + 1)];
    __shared__ float ldv[64 * 3];
    const uint32_t lane = threadIdx.x;
    if (t.abort_flag != nullptr && *t.abort_flag != 0) return; // (uniform) speculative step that did not fit its buffers: no update, no loss - the host runs it again
    if (t.loss_slots != nullptr && blockIdx.x == 0) { // *loss = the fused MSE: raster_finish_adam_kernel's fold (four wavefront sums of 64 slots, then (w0 + w1) + (w2 + w3))
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = t.loss_slots[64 * j + lane];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            w[j] = v;
        }
        if (lane == 0) *t.loss = (w[0] + w[1]) + (w[2] + w[3]);
    }
    const uint32_t N = t.N, K = t.K, KK = K - 1;
    const int degree = t.degree, Kd = (degree + 1) * (degree + 1);
    const uint32_t g0 = blockIdx.x * 64u, gmine = g0 + lane;
    const bool live = gmine < N;
    const uint32_t g = live ? gmine : (N - 1u);   // (tail lanes read the last row: no branch around the loads; they store nothing)
    constexpr int GPI = 64 / LPG;
    const int k = lane % LPG;
    // ---- phase 1 ------------------------------------------------------------------------------------------------------------------------------------------
    bool on = false;
    f3 d{0.f, 0.f, 0.f};
    float inorm = 1.f;
    const float4* a4 = reinterpret_cast<const float4*>(t.acc + size_t(g) * ACC_STRIDE);
    const float4 a3 = a4[3];
    const f3 mu = ld3(t.means, g);
    {
#pragma clang fp contract(off)
        const int2 rr = *reinterpret_cast<const int2*>(t.radii + 2 * size_t(g));
        const f3 kc = ld3(t.colors, g);
        const f3 cp = campos_of(t.viewmat);
        float b[25];
#pragma unroll
        for (int kk = 0; kk < 25; ++kk) b[kk] = 0.f;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        on = live && rr.x > 0 && rr.y > 0;
        if (on) {
            d = {mu.x - cp.x, mu.y - cp.y, mu.z - cp.z};
            if (degree >= 1) { inorm = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d = {d.x * inorm, d.y * inorm, d.z * inorm}; }
            sh_basis<false, (LPG > 16 ? 4 : 3)>(degree, d.x, d.y, d.z, b, nullptr, nullptr, nullptr);
            v0 = (kc.x > 0.f) ? a3.y : 0.f; v1 = (kc.y > 0.f) ? a3.z : 0.f; v2 = (kc.z > 0.f) ? a3.w : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < LPG; ++kk) lds_b[lane * (LPG + 1) + kk] = (kk < 25) ? b[kk] : 0.f;
        ldv[lane * 3] = v0; ldv[lane * 3 + 1] = v1; ldv[lane * 3 + 2] = v2;
    }
    // the finish pass's operands (raster_finish_adam_kernel, one round trip): in flight under phase 2
    const uint32_t o4 = g << 2, o12 = g * 12u, o16 = g << 4;
    auto at = [](const void* base, uint32_t byte_off) { return reinterpret_cast<const char*>(base) + byte_off; };
    auto ld3o = [&](const float* base, float (&dst)[3]) { const V3f x = *reinterpret_cast<const V3f*>(at(base, o12)); dst[0] = x.a[0]; dst[1] = x.a[1]; dst[2] = x.a[2]; };
    auto ld4o = [&](const float* base) { return *reinterpret_cast<const float4*>(at(base, o16)); };
    auto ld1o = [&](const float* base) { return *reinterpret_cast<const float*>(at(base, o4)); };
    const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2];
    float sc[3], m0[3], v0m[3], p1[3], m1[3], v1[3];
    ld3o(t.scales, sc);
    const float4 q = ld4o(t.quats), rq = ld4o(t.raw_quats);
    const float o = ld1o(t.opacities);
    const FinishAdam& ad = t.fin;
    ld3o(ad.m[0], m0); ld3o(ad.v[0], v0m); ld3o(t.raw_scales, p1); ld3o(ad.m[1], m1); ld3o(ad.v[1], v1);
    float4 mq = ld4o(ad.m[2]), vq4 = ld4o(ad.v[2]);
    float po = ld1o(t.raw_opacities), mo = ld1o(ad.m[3]), vo = ld1o(ad.v[3]);
    __syncthreads();
    // ---- phase 2: s_k ------------------------------------------------------------------------------------------------------------------------------------
    const uint32_t lane_el = ((lane / LPG) * KK + uint32_t(k - 1)) * 3u;   // (k == 0 lanes never use it)
    // phase 5's row addressing (the same lane meets the same rows in phase 2: LFS_TAIL_KEEP)
    const bool row_k = uint32_t(k) < K;
    float* const pbase = (k == 0) ? t.sh0 : t.shN;
    float* const mbase = (k == 0) ? t.m0 : t.mN;
    float* const vbase = (k == 0) ? t.v0 : t.vN;
    auto row_el = [&](const int it) -> size_t {
        const size_t gg = size_t(g0) + uint32_t(it) * GPI + lane / LPG;
        return (k == 0) ? gg * 3 : (gg * KK + uint32_t(k - 1)) * 3;
    };
    auto row_ok = [&](const int it) { return row_k && (g0 + uint32_t(it) * GPI + lane / LPG) < N; };
#if LFS_TAIL_KEEP
    V3f PK[LPG];
#pragma unroll
    for (int it = 0; it < LPG; ++it) {
        PK[it].a[0] = PK[it].a[1] = PK[it].a[2] = 0.f;
        if (row_ok(it)) PK[it] = *reinterpret_cast<const V3f*>(pbase + row_el(it));
    }
#endif
    {
#pragma clang fp contract(off)
#if LFS_TAIL_KEEP
#pragma unroll
#else
#pragma unroll 4
#endif
        for (int it = 0; it < LPG; ++it) {
            const uint32_t gl = it * GPI + lane / LPG;
            const float w0 = ldv[gl * 3], w1 = ldv[gl * 3 + 1], w2 = ldv[gl * 3 + 2];
            float sk = 0.f;
            // a zero gradient gives 0 x (finite) = 0 whatever the row holds: its 12 bytes are not fetched (sh_pipe_dirs_kernel, sh.hip)
            if (g0 + gl < N && k >= 1 && k < Kd && uint32_t(k) < K && (w0 != 0.f || w1 != 0.f || w2 != 0.f)) {
#if LFS_TAIL_KEEP
                const V3f p = PK[it];
#else
                const V3f p = *reinterpret_cast<const V3f*>(t.shN + size_t(g0 + uint32_t(it) * GPI) * KK * 3u + lane_el);
#endif
                sk = p.a[0] * w0 + p.a[1] * w1 + p.a[2] * w2;
            }
            lds_s[gl * (LPG + 1) + k] = sk;
        }
    }
    // phase 5's moment rows: LFS_TAIL_DEPTH of them per lane in flight
    constexpr int D = (LPG < LFS_TAIL_DEPTH) ? LPG : LFS_TAIL_DEPTH;
#if !LFS_TAIL_KEEP
    V3f P[D];
#endif
    V3f M[D], Q[D];
    auto load = [&](const int it, const int slot) {
#if !LFS_TAIL_KEEP
        P[slot].a[0] = P[slot].a[1] = P[slot].a[2] = 0.f;
#endif
        if (row_ok(it)) {
            const size_t e = row_el(it);
#if !LFS_TAIL_KEEP
            P[slot] = *reinterpret_cast<const V3f*>(pbase + e);
#endif
            M[slot] = *reinterpret_cast<const V3f*>(mbase + e); Q[slot] = *reinterpret_cast<const V3f*>(vbase + e);
        }
    };
#if LFS_TAIL_EARLY
#pragma unroll
    for (int it = 0; it < D; ++it) load(it, it);
#endif
    __syncthreads();
    // ---- phase 3: dL/d(dirs), then the finish pass ------------------------------------------------------------------------------------------------------------
    float vd[3] = {0.f, 0.f, 0.f};
    if (on && degree >= 1) {
#pragma clang fp contract(off)
        float b[25], bx[25], by[25], bz[25];
        sh_basis<true, (LPG > 16 ? 4 : 3)>(degree, d.x, d.y, d.z, b, bx, by, bz);
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int kk = 1; kk < LPG && kk < 25; ++kk) {
            const float sk = lds_s[lane * (LPG + 1) + kk];
            gx += bx[kk] * sk; gy += by[kk] * sk; gz += bz[kk] * sk;
        }
        const float dd = gx * d.x + gy * d.y + gz * d.z;
        vd[0] = (gx - dd * d.x) * inorm; vd[1] = (gy - dd * d.y) * inorm; vd[2] = (gz - dd * d.z) * inorm;
    }
    float pm[3] = {mu.x, mu.y, mu.z};   // the mean: updated below, read again by phase 4
    {   // raster_finish_adam_kernel<true> from its "everything is in flight" line on, statement by statement
        const float A[9] = {a0.x * REC_UNSCALE, a0.y * REC_UNSCALE, a0.z * REC_UNSCALE, a0.w * REC_UNSCALE, a1.x * REC_UNSCALE, a1.y * REC_UNSCALE, a1.z * REC_UNSCALE,
                            a1.w * REC_UNSCALE, a2.x * REC_UNSCALE};
        const f3 G{-a2.y * REC_UNSCALE, -a2.z * REC_UNSCALE, -a2.w * REC_UNSCALE};
        bool any = G.x != 0.f || G.y != 0.f || G.z != 0.f;
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) any |= A[kk] != 0.f;
        const float v_opac = a3.x != 0.f ? a3.x / o : 0.f;
        float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
        {
            const float is[3] = {1.f / sc[0], 1.f / sc[1], 1.f / sc[2]};
            finish_geometry<true>(q, is, A, G, mu, cams[0], vm, vq, vs);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) { vm[kk] = any ? vm[kk] : 0.f; vs[kk] = any ? vs[kk] : 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) vq[kk] = any ? vq[kk] : 0.f;
        }
        {
#pragma clang fp contract(off)
            float gm[3] = {vm[0] + vd[0], vm[1] + vd[1], vm[2] + vd[2]};
            const float nrm = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
            float gq[4];
            if (nrm > 1e-12f) {
                const float inv = 1.f / nrm;
                const float y[4] = {rq.x * inv, rq.y * inv, rq.z * inv, rq.w * inv};
                const float dq = vq[0] * y[0] + vq[1] * y[1] + vq[2] * y[2] + vq[3] * y[3];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) gq[kk] = (vq[kk] - dq * y[kk]) * inv;
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) gq[kk] = vq[kk] * 1e12f;
            }
            float gs[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) gs[kk] = (vs[kk] + ad.scale_reg) * sc[kk];
            const float go = (v_opac + ad.opacity_reg) * o * (1.f - o);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                adam_elem(pm[kk], m0[kk], v0m[kk], gm[kk], ad.s[0]);
                adam_elem(p1[kk], m1[kk], v1[kk], gs[kk], ad.s[1]);
            }
            float pq[4] = {rq.x, rq.y, rq.z, rq.w};
            adam_elem(pq[0], mq.x, vq4.x, gq[0], ad.s[2]); adam_elem(pq[1], mq.y, vq4.y, gq[1], ad.s[2]);
            adam_elem(pq[2], mq.z, vq4.z, gq[2], ad.s[2]); adam_elem(pq[3], mq.w, vq4.w, gq[3], ad.s[2]);
            adam_elem(po, mo, vo, go, ad.s[3]);
            if (live) {
                auto st3a = [&](float* base, const float (&src)[3]) { V3f x; x.a[0] = src[0]; x.a[1] = src[1]; x.a[2] = src[2]; *reinterpret_cast<V3f*>(const_cast<char*>(at(base, o12))) = x; };
                st3a(t.means, pm); st3a(ad.m[0], m0); st3a(ad.v[0], v0m);
                st3a(t.raw_scales, p1); st3a(ad.m[1], m1); st3a(ad.v[1], v1);
                reinterpret_cast<float4*>(t.raw_quats)[gmine] = make_float4(pq[0], pq[1], pq[2], pq[3]);
                reinterpret_cast<float4*>(ad.m[2])[gmine] = mq; reinterpret_cast<float4*>(ad.v[2])[gmine] = vq4;
                t.raw_opacities[gmine] = po; ad.m[3][gmine] = mo; ad.v[3][gmine] = vo;
            }
        }
    }
    // ---- phase 4: the next view's basis (every Gaussian: its visibility there is not known yet) -> lds_s (this lane's own row: phase 3 has consumed it) -----------------
    if (NEXT) {
#pragma clang fp contract(off)
        const f3 cp = campos_of(t.next_viewmat);
        float b[25];
#pragma unroll
        for (int kk = 0; kk < 25; ++kk) b[kk] = 0.f;
        if (live) {   // sh_fwd_kernel<LPG, true> with radii == NULL
            f3 dn{pm[0] - cp.x, pm[1] - cp.y, pm[2] - cp.z};
            if (degree >= 1) { const float inn = 1.f / sqrtf(dn.x * dn.x + dn.y * dn.y + dn.z * dn.z); dn = {dn.x * inn, dn.y * inn, dn.z * inn}; }
            sh_basis<false, (LPG > 16 ? 4 : 3)>(degree, dn.x, dn.y, dn.z, b, nullptr, nullptr, nullptr);
        }
#pragma unroll
        for (int kk = 0; kk < LPG; ++kk) lds_s[lane * (LPG + 1) + kk] = (kk < 25) ? b[kk] : 0.f;
    }
    __syncthreads();
    // ---- phase 5: Adam on the coefficient rows (+ the next colours) ------------------------------------------------------------------------------------------------
    {
#pragma clang fp contract(off)
        const AdamScalars as = (k == 0) ? t.s0 : t.sN;
        auto st_mom = [&](float* q, const V3f& x) {
#if LFS_TAIL_NT
            __builtin_nontemporal_store(x.a[0], q); __builtin_nontemporal_store(x.a[1], q + 1); __builtin_nontemporal_store(x.a[2], q + 2);
#else
            *reinterpret_cast<V3f*>(q) = x;
#endif
        };
#if !LFS_TAIL_EARLY
#pragma unroll
        for (int it = 0; it < D; ++it) load(it, it);
#endif
#if LFS_TAIL_KEEP
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int it0 = 0; it0 < LPG; it0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int it = it0 + u;
                const uint32_t gl = it * GPI + lane / LPG;
#if LFS_TAIL_KEEP
                V3f p = PK[it];
#else
                V3f p = P[u];
#endif
                if (row_ok(it)) {
                    const float bk = lds_b[gl * (LPG + 1) + k];
                    const float o0 = bk * ldv[gl * 3], o1 = bk * ldv[gl * 3 + 1], o2 = bk * ldv[gl * 3 + 2];
                    V3f m = M[u], qq = Q[u];
                    adam_elem(p.a[0], m.a[0], qq.a[0], o0, as); adam_elem(p.a[1], m.a[1], qq.a[1], o1, as); adam_elem(p.a[2], m.a[2], qq.a[2], o2, as);
                    const size_t e = row_el(it);
                    *reinterpret_cast<V3f*>(pbase + e) = p; st_mom(mbase + e, m); st_mom(vbase + e, qq);
                }
                if (NEXT) {   // sh_fwd_kernel's phase 2 on the updated row (rows k >= Kd meet a zero basis value; lanes without a row contribute c = 0)
                    const float bn = lds_s[gl * (LPG + 1) + k];
                    const bool use = row_ok(it) && k < Kd;
                    float r0 = bn * (use ? p.a[0] : 0.f), r1 = bn * (use ? p.a[1] : 0.f), r2 = bn * (use ? p.a[2] : 0.f);
                    r0 = group_sum<LPG>(r0); r1 = group_sum<LPG>(r1); r2 = group_sum<LPG>(r2);
                    if (k == 0 && g0 + gl < N) {
                        float* co = t.colors + 3 * size_t(g0 + gl);
                        co[0] = fmaxf(r0 + 0.5f, 0.f); co[1] = fmaxf(r1 + 0.5f, 0.f); co[2] = fmaxf(r2 + 0.5f, 0.f);
                    }
                }
                if (it + D < LPG) load(it + D, u);
            }
        }
    }
}

// deterministic mode, after pass 2: acc[i] = fixed-point sum * 2^(e - 40) (e from the pass-1 maximum that acc[i] still holds as bits)
__global__ void __launch_bounds__(256) raster_det_resolve_kernel(const size_t n, float* __restrict__ acc, const unsigned long long* __restrict__ det64) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t mbits = reinterpret_cast<const uint32_t*>(acc)[i];
    float out = 0.f;
    if (mbits != 0u) {
        const int e = max(int((mbits >> 23) & 0xffu), 1) - 127;
        out = float(ldexp(double((long long)det64[i]), e - 40)); // exact scaling; one rounding to float
    }
    acc[i] = out;
}

// Launch geometry of cull / fwd / bwd: one wavefront per 8x8 cell, whole workgroups per tile.
struct RasterGeom { uint32_t tw, th, blocks_per_tile, waves_per_block, threads, grid, wpt; uint64_t cells; };
static uint32_t g_debug_flags = 0; // bit 0: keep every tile-list entry in the cell lists (no culling: the bit-identity tests); bit 4: deterministic backward accumulation
                                   // (two passes, 64-bit fixed point; 3 channels); bit 5: one-pass intersection scatter (intersect.hip). Bits 1-3 were the removed experiments.
static bool raster_geom(const lfs_cameras* cams, uint32_t tile_size, RasterGeom& g) {
    if (tile_size < 8 || tile_size > 64 || (tile_size & 7)) return false;
    g.tw = (cams->image_width + tile_size - 1) / tile_size;
    g.th = (cams->image_height + tile_size - 1) / tile_size;
    g.wpt = (tile_size / 8) * (tile_size / 8);
    g.waves_per_block = (g.wpt % 4 == 0) ? 4 : (g.wpt % 2 == 0) ? 2 : 1; // whole workgroups per tile (9, 25, 49 cells: one wave each)
    g.blocks_per_tile = g.wpt / g.waves_per_block;
    g.threads = g.waves_per_block * 64;
    const uint64_t nb = uint64_t(cams->C) * g.tw * g.th * g.blocks_per_tile;
    g.grid = cell_grid_blocks(nb, g.blocks_per_tile);
    g.cells = uint64_t(cams->C) * g.tw * g.th * g.wpt;
    return true;
}
// fwd / bwd never cooperate across the wavefronts of a workgroup (only the cull kernel shares its gathers through LDS): with one wavefront per workgroup a
// finished cell frees its slot at once instead of waiting for the slowest of its tile's four
static RasterGeom wave_geom(const lfs_cameras* cams, const RasterGeom& g) {
    RasterGeom w = g;
#if LFS_RASTER_WAVE_BLOCKS
    w.waves_per_block = 1; w.blocks_per_tile = g.wpt; w.threads = 64;
    const uint64_t nb = uint64_t(cams->C) * g.tw * g.th * w.blocks_per_tile;
    w.grid = cell_grid_blocks(nb, w.blocks_per_tile);
#else
    (void)cams;
#endif
    return w;
}

} // namespace lfs

using namespace lfs;

extern "C" void lfs_set_debug_flags(uint32_t flags) { g_debug_flags = flags; }
extern "C" uint32_t lfs_get_debug_flags(void) { return g_debug_flags; }

extern "C" size_t lfs_rasterize_workspace_bytes(uint32_t C, uint32_t N, uint32_t channels, uint32_t image_width, uint32_t image_height,
                                                uint32_t tile_size, int64_t n_isects) {
    (void)channels;
    lfs_cameras cams{};
    cams.C = C; cams.image_width = image_width; cams.image_height = image_height;
    RasterGeom g;
    if (!raster_geom(&cams, tile_size, g) || n_isects < 0) return 0;
    return raster_ws(nullptr, C, N, g.cells, uint64_t(g.wpt) * uint64_t(n_isects), (g_debug_flags & 16u) != 0).bytes;
}
size_t lfs::raster_workspace_bytes_for(uint32_t N, uint32_t image_width, uint32_t image_height, uint32_t tile_size, int64_t capacity) {
    return lfs_rasterize_workspace_bytes(1, N, 3, image_width, image_height, tile_size, capacity);
}

static int raster_mode(const lfs_cameras* cams) {
    if (cams->rs_type != LFS_SHUTTER_GLOBAL) return lfs::RAY_ROLLING;
    return lfs::RAY_GLOBAL;
}

static int raster_check(uint32_t N, uint32_t channels, const lfs_cameras* cams, uint32_t tile_size, RasterGeom& g) {
    if (!cams || !cams->viewmats0 || !cams->Ks || cams->C == 0) return LFS_E_INVALID;
    if (cams->camera_model != LFS_CAMERA_PINHOLE && cams->camera_model != LFS_CAMERA_FISHEYE) return LFS_E_UNSUPPORTED;
    if (channels < 1 || channels > 4) return LFS_E_UNSUPPORTED; // Rasterization.cpp:65 asserts 3; depth modes need 1 and 4
    if (!raster_geom(cams, tile_size, g)) return LFS_E_UNSUPPORTED;
    if (uint64_t(cams->C) * N >= (1ull << 26)) return LFS_E_UNSUPPORTED; // 32-bit byte offsets of the record walker (4 GB of 64-B records)
#if LFS_RED_BUF_ATOMIC
    if (uint64_t(cams->C) * N >= (1ull << 25)) return LFS_E_UNSUPPORTED; // the backward's buffer atomic: accumulator rows below RED_BUF_DEAD = 2 GB (lfs_raster_common.cuh)
#endif
    return LFS_OK;
}

// n_isects >= 0: the host knows the count (operator calls); the workspace holds lists for exactly that many intersections.
// n_isects < 0 (guarded training step, lfs_step_internal.h): the count is on the device - tile_offsets has T + 1 entries, the kernels read the last one -
// and the lists are sized for `capacity`.
struct IsectCount { int64_t n_isects, capacity; int32_t arg() const { return n_isects >= 0 ? int32_t(n_isects) : -1; } int64_t sized() const { return n_isects >= 0 ? n_isects : capacity; } };

// camera state, 64-byte records + culling records, compacted per-cell lists: everything fwd and bwd walk
static void raster_prepare(const RasterWs& w, const RasterGeom& g, uint32_t N, uint32_t channels, const float* means, const float* quats,
                           const float* scales, const float* colors, const float* opacities, const uint8_t* masks,
                           const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                           const IsectCount ic, hipStream_t s, bool cams_ready = false, bool records_ready = false) {
    const uint32_t C = cams->C;
    const bool uniform = cams->rs_type == LFS_SHUTTER_GLOBAL;
    if (!cams_ready) hipLaunchKernelGGL(cam_prep_kernel, dim3(1), dim3(64), 0, s, *cams, w.cams); // (training step: written by the projection kernel already)
    const size_t CN = size_t(C) * N;
    if (CN > 0 && !records_ready) { // (training step, round 4: records + culling records written by the projection kernel - projection_ut.hip, PACK)
        lfs::ProfScope prof_pack("raster_pack", s);
        const dim3 pg(uint32_t((CN + 255) / 256));
        if (uniform) hipLaunchKernelGGL(raster_pack_kernel<true>, pg, dim3(256), 0, s, C, N, channels, means, quats, scales, colors, opacities, w.cams, w.recs, w.cull);
        else hipLaunchKernelGGL(raster_pack_kernel<false>, pg, dim3(256), 0, s, C, N, channels, means, quats, scales, colors, opacities, w.cams, w.recs, w.cull);
    }
    lfs::ProfScope prof_cull("raster_cull", s);
    const uint32_t cull_on = (g_debug_flags & 1u) ? 0u : 1u;
#define LFS_CULL(U)                                                                                                                     \
    hipLaunchKernelGGL((raster_cull_kernel<U>), dim3(g.grid), dim3(g.threads), 0, s, C, g.tw, g.th, cams->image_width, cams->image_height, \
                       tile_size, g.blocks_per_tile, g.waves_per_block, cull_on, w.cams, w.cull, masks, tile_offsets, flatten_ids,        \
                       ic.arg(), w.cell_count, w.cell_list)
    if (uniform) LFS_CULL(true); else LFS_CULL(false);
#undef LFS_CULL
}

static int raster_fwd_impl(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const IsectCount ic,
    float* render_colors, float* render_alphas, int32_t* last_ids,
    void* workspace, size_t workspace_bytes, hipStream_t s, bool cams_ready = false, bool records_ready = false, hipEvent_t wait_before_fwd = nullptr) {
    RasterGeom g;
    int rc = raster_check(N, channels, cams, tile_size, g);
    if (rc) return rc;
    if (!render_colors || !render_alphas || !last_ids || !tile_offsets || !workspace) return LFS_E_INVALID;
    const int64_t n_sized = ic.sized();
    if (n_sized < 0 || n_sized > 0x7FFFFFFFll) return LFS_E_INVALID;
    if (uint64_t(n_sized) >= (1ull << 29)) return LFS_E_UNSUPPORTED; // 32-bit byte offsets inside one cell list
    const uint32_t C = cams->C;
    RasterWs w = raster_ws(workspace, C, N, g.cells, uint64_t(g.wpt) * uint64_t(n_sized), (g_debug_flags & 16u) != 0);
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    if (N > 0 && (!means || !quats || !scales || !colors || !opacities)) return LFS_E_INVALID;
    if (n_sized > 0 && !flatten_ids) return LFS_E_INVALID;
    raster_prepare(w, g, N, channels, means, quats, scales, colors, opacities, masks, cams, tile_size, tile_offsets, flatten_ids, ic, s, cams_ready, records_ready);
    // pipelined step: the culling above needs geometry only; the colours of the records arrive from the side stream (gut_step.hip)
    if (wait_before_fwd != nullptr) { const hipError_t e = hipStreamWaitEvent(s, wait_before_fwd, 0); if (e != hipSuccess) return (int)e; }
    lfs::ProfScope prof("raster_fwd", s);
    const RasterGeom gw = wave_geom(cams, g);
#define LFS_FWD(CD, MODE)                                                                                        \
    hipLaunchKernelGGL((raster_fwd_kernel<CD, MODE>), dim3(gw.grid), dim3(gw.threads), 0, s, C, N, g.tw, g.th,     \
                       cams->image_width, cams->image_height, tile_size, gw.blocks_per_tile, gw.waves_per_block, \
                       w.cams, w.recs, colors, backgrounds, masks, tile_offsets, w.cell_count, w.cell_list, ic.arg(), \
                       render_colors, render_alphas, last_ids, reinterpret_cast<int32_t*>(w.cell_list))
    switch (channels * 2 + raster_mode(cams)) {
    case 2: LFS_FWD(1, 0); break; case 3: LFS_FWD(1, 1); break;
    case 4: LFS_FWD(2, 0); break; case 5: LFS_FWD(2, 1); break;
    case 6: LFS_FWD(3, 0); break; case 7: LFS_FWD(3, 1); break;
    case 8: LFS_FWD(4, 0); break; default: LFS_FWD(4, 1); break;
    }
#undef LFS_FWD
    return (int)hipGetLastError();
}

extern "C" int lfs_rasterize_to_pixels_from_world_3dgs_fwd(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size, const lfs_ut_params* ut_params,
    const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t n_isects,
    float* render_colors, float* render_alphas, int32_t* last_ids,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    (void)ut_params; // carried by the reference signature, unused by its rasterizer as well
    if (n_isects < 0) return LFS_E_INVALID;
    return raster_fwd_impl(N, channels, means, quats, scales, colors, opacities, backgrounds, masks, cams, tile_size, tile_offsets, flatten_ids, IsectCount{n_isects, 0},
                           render_colors, render_alphas, last_ids, workspace, workspace_bytes, (hipStream_t)stream);
}

int lfs::raster_fwd_guarded(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                            const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                            int64_t capacity, float* render_colors, float* render_alphas, int32_t* last_ids, void* workspace, size_t workspace_bytes, hipStream_t s,
                            bool cams_ready, bool records_ready, hipEvent_t wait_before_fwd) {
    if (capacity < 0) return LFS_E_INVALID;
    return raster_fwd_impl(N, 3, means, quats, scales, colors, opacities, backgrounds, nullptr, cams, tile_size, tile_offsets, flatten_ids, IsectCount{-1, capacity},
                           render_colors, render_alphas, last_ids, workspace, workspace_bytes, s, cams_ready, records_ready, wait_before_fwd);
}

// where the camera state / records / culling records of a (one-camera) rasterizer workspace live: the training step's projection kernel writes them there
void lfs::raster_workspace_parts(void* workspace, uint32_t N, void** cams_dev, void** recs, void** cull) {
    const RasterWs w = raster_ws(workspace, 1, N, 0, 0);
    if (cams_dev) *cams_dev = w.cams;
    if (recs) *recs = w.recs;
    if (cull) *cull = w.cull;
}

static int raster_bwd_impl(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const IsectCount ic,
    const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas,
    float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities,
    void* workspace, size_t workspace_bytes, hipStream_t s, bool prepared, const MseFuse* mse = nullptr, bool finish = true) {
    RasterGeom g;
    int rc = raster_check(N, channels, cams, tile_size, g);
    if (rc) return rc;
    if (!workspace || !tile_offsets) return LFS_E_INVALID;
    if (mse && (channels != 3 || cams->C != 1 || masks || v_render_alphas || !mse->render || !mse->target || !mse->loss)) return LFS_E_INVALID;
    const int64_t n_sized = ic.sized();
    if (n_sized < 0 || n_sized > 0x7FFFFFFFll) return LFS_E_INVALID;
    if (uint64_t(n_sized) >= (1ull << 29)) return LFS_E_UNSUPPORTED; // 32-bit byte offsets inside one cell list
    const uint32_t C = cams->C;
    const bool det = (g_debug_flags & 16u) != 0 && channels == 3;
    RasterWs w = raster_ws(workspace, C, N, g.cells, uint64_t(g.wpt) * uint64_t(n_sized), (g_debug_flags & 16u) != 0);
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    if (N == 0) return LFS_OK;
    if (!means || !quats || !scales || !colors || !opacities) return LFS_E_INVALID;
    if (finish && (!v_means || !v_quats || !v_scales || !v_colors || !v_opacities)) return LFS_E_INVALID;
    if (!finish && (channels != 3 || C != 1)) return LFS_E_INVALID; // the accumulator-only forms feed lfs_gut_finish_adam / lfs_gut_finish_grads (one camera, RGB)
    if (n_sized > 0 && (!flatten_ids || !render_alphas || !last_ids || (!v_render_colors && !mse))) return LFS_E_INVALID; // v_render_alphas == NULL: zeros
    const bool uniform = cams->rs_type == LFS_SHUTTER_GLOBAL;
    const size_t CN = size_t(C) * N;
    hipError_t e = hipMemsetAsync(w.acc, 0, sizeof(float) * (ACC_STRIDE * CN + (mse ? LOSS_SLOTS : 0)), s);
    MseFuse mse_dev{};
    if (mse) { mse_dev = *mse; mse_dev.loss = w.acc + ACC_STRIDE * CN; } // the kernel adds into the slots, raster_finish folds them into *loss
    if (e != hipSuccess) return (int)e;
    if (det) { e = hipMemsetAsync(w.det64, 0, sizeof(unsigned long long) * ACC_STRIDE * CN, s); if (e != hipSuccess) return (int)e; mse_dev.det64 = w.det64; }
    if (channels > 3) { e = hipMemsetAsync(v_colors, 0, sizeof(float) * channels * CN, s); if (e != hipSuccess) return (int)e; }
    // self-contained call: camera state, records and cell lists are rebuilt; "prepared" = the caller guarantees
    // the workspace still holds what the forward call with the same inputs left there
    if (!prepared) raster_prepare(w, g, N, channels, means, quats, scales, colors, opacities, masks, cams, tile_size, tile_offsets, flatten_ids, ic, s);
    if (n_sized > 0) {
        // One launch (every mode but the deterministic one): timed, when asked for, through the dispatch packet's own signal (lfs_prof.h, prof_kernel_events) - no
        // event-record packets around the dominant kernel inside bench.py's timed region. The deterministic mode's two passes + resolve keep the event scope.
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
#ifndef LFS_EMULATE
        const bool ext_timed = !det && LFS_PROF_EXT_LAUNCH && lfs::prof_kernel_events("raster_bwd", &pe0, &pe1);
#else
        const bool ext_timed = false;
#endif
        lfs::ProfScope prof(ext_timed ? "" : "raster_bwd", s);
        const RasterGeom gw = wave_geom(cams, g);
#ifndef LFS_EMULATE
#define LFS_BWD_LAUNCH(KERNEL, ...) do { if (ext_timed) hipExtLaunchKernelGGL(KERNEL, dim3(gw.grid), dim3(gw.threads), 0, s, pe0, pe1, 0, __VA_ARGS__); \
                                         else hipLaunchKernelGGL(KERNEL, dim3(gw.grid), dim3(gw.threads), 0, s, __VA_ARGS__); } while (0)
#else
#define LFS_BWD_LAUNCH(KERNEL, ...) hipLaunchKernelGGL(KERNEL, dim3(gw.grid), dim3(gw.threads), 0, s, __VA_ARGS__)
#endif
#define LFS_BWD_K(CD, MODE, ...)                                                                                 \
    LFS_BWD_LAUNCH((raster_bwd_kernel<CD, MODE, ##__VA_ARGS__>), C, N, g.tw, g.th, \
                       cams->image_width, cams->image_height, tile_size, gw.blocks_per_tile, gw.waves_per_block, \
                       w.cams, w.recs, colors, backgrounds, masks, tile_offsets, w.cell_count, w.cell_list, ic.arg(), \
                       render_alphas, last_ids, v_render_colors, v_render_alphas, w.acc, v_colors, mse_dev)
        if (det) { // pass 1: per-slot maxima of |total| (integer atomicMax), pass 2: 64-bit fixed-point sums, then back to float
#define LFS_BWD_DET(MODE, LOSSV) do { LFS_BWD_K(3, MODE, LOSSV, 1); LFS_BWD_K(3, MODE, LOSSV, 2); } while (0)
            if (mse) { if (raster_mode(cams) == 0) LFS_BWD_DET(0, true); else LFS_BWD_DET(1, true); }
            else { if (raster_mode(cams) == 0) LFS_BWD_DET(0, false); else LFS_BWD_DET(1, false); }
#undef LFS_BWD_DET
            const size_t n_acc = ACC_STRIDE * CN;
            hipLaunchKernelGGL(raster_det_resolve_kernel, dim3(uint32_t((n_acc + 255) / 256)), dim3(256), 0, s, n_acc, w.acc, w.det64);
        } else if (mse) {
            if (raster_mode(cams) == 0) LFS_BWD_K(3, 0, true); else LFS_BWD_K(3, 1, true);
        } else
        switch (channels * 2 + raster_mode(cams)) {
        case 2: LFS_BWD_K(1, 0); break; case 3: LFS_BWD_K(1, 1); break;
        case 4: LFS_BWD_K(2, 0); break; case 5: LFS_BWD_K(2, 1); break;
        case 6: LFS_BWD_K(3, 0); break; case 7: LFS_BWD_K(3, 1); break;
        case 8: LFS_BWD_K(4, 0); break; default: LFS_BWD_K(4, 1); break;
        }
#undef LFS_BWD_K
#undef LFS_BWD_LAUNCH
    }
    if (!finish) return (int)hipGetLastError();
    const dim3 fg((N + 255) / 256);
    lfs::ProfScope prof_fin("raster_finish", s);
    const float* slots = mse ? w.acc + ACC_STRIDE * CN : nullptr;
    float* loss_out = mse ? mse->loss : nullptr;
    if (uniform) hipLaunchKernelGGL(raster_finish_kernel<true>, fg, dim3(256), 0, s, C, N, channels, means, quats, scales, w.cams, w.acc, v_means, v_quats, v_scales, v_colors, v_opacities, slots, loss_out, opacities);
    else hipLaunchKernelGGL(raster_finish_kernel<false>, fg, dim3(256), 0, s, C, N, channels, means, quats, scales, w.cams, w.acc, v_means, v_quats, v_scales, v_colors, v_opacities, slots, loss_out, opacities);
    return (int)hipGetLastError();
}

extern "C" int lfs_rasterize_to_pixels_from_world_3dgs_bwd(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size, const lfs_ut_params* ut_params,
    const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t n_isects,
    const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas,
    float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    (void)ut_params;
    if (n_isects < 0) return LFS_E_INVALID;
    return raster_bwd_impl(N, channels, means, quats, scales, colors, opacities, backgrounds, masks, cams, tile_size, tile_offsets, flatten_ids,
                           IsectCount{n_isects, 0}, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors,
                           v_opacities, workspace, workspace_bytes, (hipStream_t)stream, false);
}

extern "C" int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size, const lfs_ut_params* ut_params,
    const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t n_isects,
    const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas,
    float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    (void)ut_params;
    if (n_isects < 0) return LFS_E_INVALID;
    return raster_bwd_impl(N, channels, means, quats, scales, colors, opacities, backgrounds, masks, cams, tile_size, tile_offsets, flatten_ids,
                           IsectCount{n_isects, 0}, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors,
                           v_opacities, workspace, workspace_bytes, (hipStream_t)stream, true);
}

// "prepared" backward with the clamped MSE loss of lfs_mse_loss_fwd_bwd folded in (extension): render_colors [H,W,3] = the forward
// output, target_chw [3,H,W]; *loss += weight * mean((clamp(render, 0, 1) - target)^2). One camera, 3 channels, no masks.
extern "C" int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
    int64_t n_isects, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw, float weight,
    float* loss, float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes,
    lfs_stream_t stream) {
    if (!cams || !render_colors || !target_chw || !loss || n_isects < 0) return LFS_E_INVALID;
    const MseFuse mse{render_colors, target_chw, weight / float(3u * cams->image_width * cams->image_height), loss, nullptr};
    if (n_isects == 0) return LFS_E_UNSUPPORTED; // nothing rendered: use lfs_mse_loss_fwd_bwd (the loss of the background image)
    return raster_bwd_impl(N, 3, means, quats, scales, colors, opacities, backgrounds, nullptr, cams, tile_size, tile_offsets, flatten_ids, IsectCount{n_isects, 0},
                           render_alphas, last_ids, nullptr, nullptr, v_means, v_quats, v_scales, v_colors, v_opacities, workspace, workspace_bytes,
                           (hipStream_t)stream, true, &mse);
}


// ---- the all-inline training step (extension; one camera, global shutter, 3 channels): accumulator-only backward + fused finish / Adam ----
// lfs_rasterize_..._bwd_prepared_mse without its last kernel: the per-Gaussian sums stay in the workspace (rows of 16 floats at
// lfs_rasterize_workspace_acc_offset) together with the loss partial sums; lfs_sh_model_bwd_adam_all reads dL/dcolour from the rows and writes
// dL/d(dirs) into them, lfs_gut_finish_adam turns them into the parameter updates.
extern "C" size_t lfs_rasterize_workspace_acc_offset(uint32_t C, uint32_t N) {
    const RasterWs w = raster_ws(nullptr, C, N, 0, 0);
    return size_t(reinterpret_cast<const char*>(w.acc) - static_cast<const char*>(nullptr));
}

static int raster_bwd_mse_acc(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                              const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                              const IsectCount ic, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw,
                              float weight, void* workspace, size_t workspace_bytes, hipStream_t s) {
    if (!cams || !render_colors || !target_chw) return LFS_E_INVALID;
    float dummy_loss;
    const MseFuse mse{render_colors, target_chw, weight / float(3u * cams->image_width * cams->image_height), &dummy_loss, nullptr}; // (.loss is redirected to the slots)
    return raster_bwd_impl(N, 3, means, quats, scales, colors, opacities, backgrounds, nullptr, cams, tile_size, tile_offsets, flatten_ids, ic,
                           render_alphas, last_ids, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, s, true, &mse, false);
}

extern "C" int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse_acc(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
    int64_t n_isects, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw, float weight,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    if (n_isects < 0) return LFS_E_INVALID;
    if (n_isects == 0) return LFS_E_UNSUPPORTED;
    return raster_bwd_mse_acc(N, means, quats, scales, colors, opacities, backgrounds, cams, tile_size, tile_offsets, flatten_ids, IsectCount{n_isects, 0}, render_colors,
                              render_alphas, last_ids, target_chw, weight, workspace, workspace_bytes, (hipStream_t)stream);
}

int lfs::raster_bwd_mse_acc_guarded(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                                    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                                    int64_t capacity, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw,
                                    float weight, void* workspace, size_t workspace_bytes, hipStream_t s) {
    if (capacity <= 0) return LFS_E_INVALID;
    return raster_bwd_mse_acc(N, means, quats, scales, colors, opacities, backgrounds, cams, tile_size, tile_offsets, flatten_ids, IsectCount{-1, capacity}, render_colors,
                              render_alphas, last_ids, target_chw, weight, workspace, workspace_bytes, s);
}

// the same for a caller-provided dL/d(render) (any loss): lfs_..._bwd_prepared without its last kernel
extern "C" int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_acc(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
    int64_t n_isects, const float* render_alphas, const int32_t* last_ids, const float* v_render_colors, const float* v_render_alphas,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    if (!cams || !v_render_colors || n_isects < 0) return LFS_E_INVALID;
    return raster_bwd_impl(N, 3, means, quats, scales, colors, opacities, backgrounds, nullptr, cams, tile_size, tile_offsets, flatten_ids, IsectCount{n_isects, 0},
                           render_alphas, last_ids, v_render_colors, v_render_alphas, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                           (hipStream_t)stream, true, nullptr, false);
}

int lfs::raster_bwd_acc_guarded(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                                const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                                int64_t capacity, const float* render_alphas, const int32_t* last_ids, const float* v_render_colors, void* workspace,
                                size_t workspace_bytes, hipStream_t s) {
    if (!cams || !v_render_colors || capacity <= 0) return LFS_E_INVALID;
    return raster_bwd_impl(N, 3, means, quats, scales, colors, opacities, backgrounds, nullptr, cams, tile_size, tile_offsets, flatten_ids, IsectCount{-1, capacity},
                           render_alphas, last_ids, v_render_colors, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, s, true, nullptr, false);
}

// The accumulator rows -> gradient TENSORS of the raw parameters in one pass (raster_finish + lfs_activations_bwd + the copy of dL/dmeans): for steps
// that need the gradients themselves (multi-GPU all-reduce, several views per step, non-MSE losses). g_* are written or, accumulate != 0, added to;
// v_colors [N,3] is written (the SH backward consumes it and adds dL/d(dirs) onto g_means). loss (nullable): += the fused MSE partial sums.
int lfs::gut_finish_grads_impl(
    uint32_t N, const float* means, const float* raw_quats, const float* quats, const float* scales, const float* opacities, float scale_reg, float opacity_reg,
    int accumulate, float* g_means, float* g_raw_scales, float* g_raw_quats, float* g_raw_opacities, float* v_colors, const float* v_dirs, float* loss,
    void* workspace, size_t workspace_bytes, hipStream_t s) {
    if (N == 0) return LFS_OK;
    if (!means || !raw_quats || !quats || !scales || !opacities || !g_means || !g_raw_scales || !g_raw_quats || !g_raw_opacities || (!v_colors && !v_dirs) || !workspace) return LFS_E_INVALID;
    const RasterWs w = raster_ws(workspace, 1, N, 0, 0);
    if (workspace_bytes < size_t(reinterpret_cast<const char*>(w.cull) - static_cast<const char*>(workspace))) return LFS_E_WORKSPACE;
    FinishAdam ad{};
    // regularisers of trainer.cpp:132-158 (as lfs_activations_bwd): scale_reg * mean(scales) over 3N values, opacity_reg * mean(opacities)
    ad.scale_reg = scale_reg / (3.f * float(N)); ad.opacity_reg = opacity_reg / float(N);
    const FinishGrads gr{g_means, g_raw_scales, g_raw_quats, g_raw_opacities, v_colors, accumulate};
    lfs::ProfScope prof("finish_grads", s);
    hipLaunchKernelGGL(raster_finish_adam_kernel<false>, dim3((N + LFS_FINISH_BLOCK - 1) / LFS_FINISH_BLOCK), dim3(LFS_FINISH_BLOCK), 0, s, N, const_cast<float*>(means), (float*)nullptr, const_cast<float*>(raw_quats),
                       (float*)nullptr, quats, scales, opacities, w.cams, w.acc, v_dirs, ad, gr, loss ? w.acc + ACC_STRIDE * size_t(N) : nullptr, loss, (const int32_t*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int lfs_gut_finish_grads(
    uint32_t N, const float* means, const float* raw_quats, const float* quats, const float* scales, const float* opacities, float scale_reg, float opacity_reg,
    int accumulate, float* g_means, float* g_raw_scales, float* g_raw_quats, float* g_raw_opacities, float* v_colors, float* loss,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    if (N != 0 && !v_colors) return LFS_E_INVALID;
    return lfs::gut_finish_grads_impl(N, means, raw_quats, quats, scales, opacities, scale_reg, opacity_reg, accumulate, g_means, g_raw_scales, g_raw_quats, g_raw_opacities,
                                      v_colors, nullptr, loss, workspace, workspace_bytes, (hipStream_t)stream);
}

// scalars[k] = {lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp} for k = means, raw_scales, raw_quats, raw_opacities; *loss = the fused MSE of the backward (stored, not added)
int lfs::gut_finish_adam_impl(
    uint32_t N, float* means, float* raw_scales, float* raw_quats, float* raw_opacities, const float* quats, const float* scales, const float* opacities,
    const float* v_dirs, float* const* exp_avg /* [4] host */, float* const* exp_avg_sq /* [4] host */, const float* scalars /* [4][6] host */, float scale_reg,
    float opacity_reg, float* loss, void* workspace, size_t workspace_bytes, hipStream_t s, const int32_t* abort_flag) {
    if (N == 0) return LFS_OK;
    if (!v_dirs) return LFS_E_INVALID;
    if (!means || !raw_scales || !raw_quats || !raw_opacities || !quats || !scales || !opacities || !exp_avg || !exp_avg_sq || !scalars || !workspace) return LFS_E_INVALID;
    const RasterWs w = raster_ws(workspace, 1, N, 0, 0);
    if (workspace_bytes < size_t(reinterpret_cast<const char*>(w.cull) - static_cast<const char*>(workspace))) return LFS_E_WORKSPACE;
    FinishAdam ad;
    for (int k = 0; k < 4; ++k) {
        if (!exp_avg[k] || !exp_avg_sq[k]) return LFS_E_INVALID;
        ad.m[k] = exp_avg[k]; ad.v[k] = exp_avg_sq[k];
        ad.s[k] = AdamScalars{scalars[6 * k], scalars[6 * k + 1], scalars[6 * k + 2], scalars[6 * k + 3], scalars[6 * k + 4], scalars[6 * k + 5]};
    }
    // regularisers of trainer.cpp:132-158 (as lfs_activations_bwd): scale_reg * mean(scales) over 3N values, opacity_reg * mean(opacities)
    ad.scale_reg = scale_reg / (3.f * float(N)); ad.opacity_reg = opacity_reg / float(N);
    lfs::ProfScope prof("finish_adam", s);
    hipLaunchKernelGGL(raster_finish_adam_kernel<true>, dim3((N + LFS_FINISH_BLOCK - 1) / LFS_FINISH_BLOCK), dim3(LFS_FINISH_BLOCK), 0, s, N, means, raw_scales, raw_quats, raw_opacities, quats, scales, opacities,
                       w.cams, w.acc, v_dirs, ad, FinishGrads{}, loss ? w.acc + ACC_STRIDE * size_t(N) : nullptr, loss, abort_flag);
    return (int)hipGetLastError();
}

// The fused tail of the all-inline step (gut_tail_kernel): SH backward + all six Adam updates (+ the next view's SH colours -> colors [N,3] when next_viewmat is given).
// LFS_E_UNSUPPORTED for K > 16 (degree 4): the caller enqueues lfs_sh_model_bwd_adam_all + lfs_gut_finish_adam instead.
int lfs::gut_tail_impl(
    uint32_t N, uint32_t K, uint32_t degrees_to_use, float* means, float* sh0, float* shN, float* raw_scales, float* raw_quats, float* raw_opacities,
    const float* quats, const float* scales, const float* opacities, const float* viewmat, const float* next_viewmat, const int32_t* radii, float* colors,
    float* const* exp_avg /* [6] host, FusedAdam group order */, float* const* exp_avg_sq, const float (*scalars)[6], float scale_reg, float opacity_reg, float* loss,
    void* workspace, size_t workspace_bytes, hipStream_t s, const int32_t* abort_flag) {
    if (N == 0) return LFS_OK;
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 3 || Kd > K || K < 2) return LFS_E_INVALID;
    if (K > 16) return LFS_E_UNSUPPORTED;
    if (!means || !sh0 || !shN || !raw_scales || !raw_quats || !raw_opacities || !quats || !scales || !opacities || !viewmat || !radii || !colors || !exp_avg || !exp_avg_sq ||
        !scalars || !workspace) return LFS_E_INVALID;
    for (int k = 0; k < 6; ++k) if (!exp_avg[k] || !exp_avg_sq[k]) return LFS_E_INVALID;
    const RasterWs w = raster_ws(workspace, 1, N, 0, 0);
    if (workspace_bytes < size_t(reinterpret_cast<const char*>(w.cull) - static_cast<const char*>(workspace))) return LFS_E_WORKSPACE;
    GutTail t{};
    t.N = N; t.K = K; t.degree = int(degrees_to_use);
    t.means = means; t.sh0 = sh0; t.shN = shN; t.raw_scales = raw_scales; t.raw_quats = raw_quats; t.raw_opacities = raw_opacities;
    t.quats = quats; t.scales = scales; t.opacities = opacities; t.viewmat = viewmat; t.next_viewmat = next_viewmat; t.radii = radii; t.colors = colors; t.acc = w.acc;
    auto sc = [&](int k) { return AdamScalars{scalars[k][0], scalars[k][1], scalars[k][2], scalars[k][3], scalars[k][4], scalars[k][5]}; };
    t.m0 = exp_avg[1]; t.v0 = exp_avg_sq[1]; t.s0 = sc(1);
    t.mN = exp_avg[2]; t.vN = exp_avg_sq[2]; t.sN = sc(2);
    const int grp[4] = {0, 3, 4, 5};   // raster_finish_adam_kernel's order: means, raw_scales, raw_quats, raw_opacities
    for (int j = 0; j < 4; ++j) { t.fin.m[j] = exp_avg[grp[j]]; t.fin.v[j] = exp_avg_sq[grp[j]]; t.fin.s[j] = sc(grp[j]); }
    t.fin.scale_reg = scale_reg / (3.f * float(N)); t.fin.opacity_reg = opacity_reg / float(N);   // (as gut_finish_adam_impl)
    t.loss_slots = loss ? w.acc + ACC_STRIDE * size_t(N) : nullptr; t.loss = loss; t.abort_flag = abort_flag;
    const dim3 grid((N + 63) / 64), block(64);
    lfs::ProfScope prof("tail_sh_finish_adam", s);
    const bool next = next_viewmat != nullptr;
    if (K <= 4) {
        if (next) hipLaunchKernelGGL((gut_tail_kernel<4, true>), grid, block, 0, s, t, w.cams); else hipLaunchKernelGGL((gut_tail_kernel<4, false>), grid, block, 0, s, t, w.cams);
    } else {
        if (next) hipLaunchKernelGGL((gut_tail_kernel<16, true>), grid, block, 0, s, t, w.cams); else hipLaunchKernelGGL((gut_tail_kernel<16, false>), grid, block, 0, s, t, w.cams);
    }
    return (int)hipGetLastError();
}

extern "C" int lfs_gut_finish_adam(
    uint32_t N, float* means, float* raw_scales, float* raw_quats, float* raw_opacities, const float* quats, const float* scales, const float* opacities,
    const float* v_dirs, float* const* exp_avg /* [4] host */, float* const* exp_avg_sq /* [4] host */, const float* scalars /* [4][6] host */, float scale_reg,
    float opacity_reg, float* loss, void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    return lfs::gut_finish_adam_impl(N, means, raw_scales, raw_quats, raw_opacities, quats, scales, opacities, v_dirs, exp_avg, exp_avg_sq, scalars, scale_reg, opacity_reg,
                                     loss, workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}

// "Row" kernels of the world-space rasterizer - a MEASURED NEGATIVE RESULT, opt-in (lfs_set_debug_flags bit 2; bench.py --row-kernels). Written at
// the end of round 1 and verified there on the CPU under the wavefront emulator (tests/test_emulated_raster.py); round 2 ran them on the GPU
// (tests/test_gpu_raster_rows.py: forward bit-identical to the default kernels, backward 1e-7 - after the DPP-hazard fix described at row_mat3
// below) and measured them SLOWER than the default 8x8-cell kernels: raster_fwd 0.258 -> 0.370 ms, raster_bwd 0.670 -> 0.757 ms, list building
// 0.068 -> 0.143 - 0.201 ms (profiles/r02/raster_rows_vs_default_pmc.txt, DESIGN.md section 6b: 1.46x fewer wave-evaluations as predicted, but 63
// VALU instructions each against 37 - the record fields arrive through DPP broadcasts instead of free SGPR operands). They stay in the tree, tested.
//
// Why: after the conic culling a SYN-B Gaussian is evaluated on 8.4 cells x 64 lanes but can composite only ~235 of those 537 pixels; with
// wave-uniform records a wavefront cannot skip the quadrants of its 8x8 cell that the Gaussian misses. Here every 16-lane DPP row owns a 4x4
// QUADRANT of the cell and walks that quadrant's own conic-culled list, so one wave-evaluation serves four different Gaussians:
//   * lists   : raster_quad_lists_kernel splits each cell list (raster_cull_kernel's output) into four quadrant lists.
//   * records : the row's 64-byte record is fetched by its 16 lanes as ONE coalesced dword each (lane l holds dword l) two evaluations
//               ahead, and a field reaches all lanes of the row through the DPP operand `row_newbcast:l` (folded into v_mul, a v_mov
//               otherwise: ~10 extra VALU per evaluation; no SGPRs, no LDS).
//   * backward: the 16 per-Gaussian sums are reduced INSIDE the row by a transpose (lane^8, half-mirror, lane^2, lane^1: 45 VALU for the
//               four Gaussians of the wave, lane l ends with the total of value l) and added with one 64-lane atomic instruction.
// SYN-B statistics (host simulation with the kernels' own conic test): sum over cells of the longest quadrant list = 5.66 M
// wave-evaluations against 8.13 M cell-list entries today (1.44x fewer); lane-level work 0.67x.
#pragma once

// lane -> (row = quadrant, pixel) inside the wavefront's 8x8 cell: row r covers x in [4 (r & 1), +4), y in [4 (r >> 1), +4)
struct RowCtx { uint32_t i, j, row, l; };
LFS_DI RowCtx row_ctx(const CellCtx& cc) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t y0 = cc.i - (lane >> 3), x0 = cc.j - (lane & 7); // the cell's origin (cell_ctx maps lane -> (lane >> 3, lane & 7))
    RowCtx r;
    r.row = lane >> 4; r.l = lane & 15;
    r.i = y0 + (r.row >> 1) * 4 + (r.l >> 2);
    r.j = x0 + (r.row & 1) * 4 + (r.l & 3);
    return r;
}

// DPP row operations (16 lanes): xor 1, xor 2 inside the quads, then the half-row mirror (l <-> 7 - l) and the row mirror (l <-> 15 - l)
template <int CTRL> LFS_DI uint32_t dpp_u32(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, true); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128, DPP_NEWBCAST0 = 0x150;
LFS_DI float row_min(float v) {
    v = fminf(v, dpp_mov<DPP_XOR1>(v)); v = fminf(v, dpp_mov<DPP_XOR2>(v));
    v = fminf(v, dpp_mov<DPP_HALF_MIRROR>(v)); return fminf(v, dpp_mov<DPP_MIRROR>(v));
}
LFS_DI float row_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v)); v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v)); return fmaxf(v, dpp_mov<DPP_MIRROR>(v));
}
LFS_DI int32_t row_max_i32(int32_t v) {
    v = max(v, int32_t(dpp_u32<DPP_XOR1>(uint32_t(v)))); v = max(v, int32_t(dpp_u32<DPP_XOR2>(uint32_t(v))));
    v = max(v, int32_t(dpp_u32<DPP_HALF_MIRROR>(uint32_t(v)))); return max(v, int32_t(dpp_u32<DPP_MIRROR>(uint32_t(v))));
}
LFS_DI float row_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v); return v + dpp_mov<DPP_MIRROR>(v);
}
LFS_DI int32_t wave_max_i32(int32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
// this lane's 16 bits of a wavefront ballot, i.e. "any / which lanes of my row"
LFS_DI uint32_t row_bits(uint64_t ballot, uint32_t row) { return uint32_t(ballot >> (row * 16u)) & 0xffffu; }

// Field F (dword index inside the 64-byte record) of the row's record: lane l of the row holds dword l, `row_newbcast:F` hands it to all 16
// lanes. DPP reads lanes of the whole row, so everything here runs in wave-converged code only (a masked-off source lane would read as 0).
// Single fields go through the builtin: the compiler folds the broadcast into a consuming v_mul / v_sub and keeps the gfx9 DPP hazard itself
// (ANY VGPR a DPP instruction reads - the permuted operand, the plain one, the accumulator of v_fmac - must have been written two wait states
// earlier by a VALU instruction). It does not fold DPP into v_fmac, so the two places with three parallel chains are asm blocks that keep the
// hazard by construction: a leading `s_nop 1` covers the inputs, inside a block an instruction only reads registers written >= 3 instructions
// earlier. (First GPU run of round 2: single-instruction asm statements without the nops were WRONG on hardware - lanes 0-3 / 8-11 of each row
// - while the emulator, which has no pipeline, agreed; tools/debug_rows.py.) Asm outputs must never feed a DPP builtin directly: the hazard
// recognizer does not look inside inline asm.
template <int F> LFS_DI float row_field(float rec) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(rec), DPP_NEWBCAST0 + F, 0xf, 0xf, true)); }
template <int F> LFS_DI float mul_field(float rec, float b) { return row_field<F>(rec) * b; }
#if defined(LFS_EMULATE) || defined(LFS_ROWS_NO_ASM) // host build of tests/emul (and a developer A/B build): the DPP operands spelled out
// (rows 0, 4, 8 of the record matrix) . x: per row the chain x.x * m0, then fma with m1, then with m2 (lfs_raster_common.cuh's fma3)
LFS_DI f3 row_mat3(float rec, const f3& x) {
    return {__builtin_fmaf(row_field<2>(rec), x.z, __builtin_fmaf(row_field<1>(rec), x.y, row_field<0>(rec) * x.x)),
            __builtin_fmaf(row_field<6>(rec), x.z, __builtin_fmaf(row_field<5>(rec), x.y, row_field<4>(rec) * x.x)),
            __builtin_fmaf(row_field<10>(rec), x.z, __builtin_fmaf(row_field<9>(rec), x.y, row_field<8>(rec) * x.x))};
}
// pix[c] = fma(colour c (fields 13..15), s, pix[c])
template <int CDIM> LFS_DI void row_color_fma(float rec, float s, float (&pix)[CDIM]) {
    pix[0] = __builtin_fmaf(row_field<13>(rec), s, pix[0]);
    if (CDIM > 1) pix[1] = __builtin_fmaf(row_field<14>(rec), s, pix[1]);
    if (CDIM > 2) pix[2] = __builtin_fmaf(row_field<15>(rec), s, pix[2]);
}
// sum_c colour c * vc[c]
template <int CDIM> LFS_DI float row_color_dot(float rec, const float (&vc)[CDIM]) {
    float cv = row_field<13>(rec) * vc[0];
    if (CDIM > 1) cv += row_field<14>(rec) * vc[1];
    if (CDIM > 2) cv += row_field<15>(rec) * vc[2];
    return cv;
}
#else
#define LFS_NB(F) " row_newbcast:" #F " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
LFS_DI f3 row_mat3(float rec, const f3& x) {
    float q0, q1, q2;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %3, %4" LFS_NB(0) "v_mul_f32_dpp %1, %3, %4" LFS_NB(4) "v_mul_f32_dpp %2, %3, %4" LFS_NB(8)
        "v_fmac_f32_dpp %0, %3, %5" LFS_NB(1) "v_fmac_f32_dpp %1, %3, %5" LFS_NB(5) "v_fmac_f32_dpp %2, %3, %5" LFS_NB(9)
        "v_fmac_f32_dpp %0, %3, %6" LFS_NB(2) "v_fmac_f32_dpp %1, %3, %6" LFS_NB(6) "v_fmac_f32_dpp %2, %3, %6" LFS_NB(10)
        : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(rec), "v"(x.x), "v"(x.y), "v"(x.z));
    return {q0, q1, q2};
}
template <int CDIM> LFS_DI void row_color_fma(float rec, float s, float (&pix)[CDIM]) {
    if constexpr (CDIM == 1) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2" LFS_NB(13) : "+v"(pix[0]) : "v"(rec), "v"(s));
    if constexpr (CDIM == 2) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %2, %3" LFS_NB(13) "v_fmac_f32_dpp %1, %2, %3" LFS_NB(14) : "+v"(pix[0]), "+v"(pix[1]) : "v"(rec), "v"(s));
    if constexpr (CDIM >= 3) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %3, %4" LFS_NB(13) "v_fmac_f32_dpp %1, %3, %4" LFS_NB(14) "v_fmac_f32_dpp %2, %3, %4" LFS_NB(15)
                                 : "+v"(pix[0]), "+v"(pix[1]), "+v"(pix[2]) : "v"(rec), "v"(s));
}
template <int CDIM> LFS_DI float row_color_dot(float rec, const float (&vc)[CDIM]) {
    if constexpr (CDIM == 1) {
        return row_field<13>(rec) * vc[0];
    } else if constexpr (CDIM == 2) {
        float p0, p1;
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %2, %3" LFS_NB(13) "v_mul_f32_dpp %1, %2, %4" LFS_NB(14) : "=&v"(p0), "=&v"(p1) : "v"(rec), "v"(vc[0]), "v"(vc[1]));
        return p0 + p1;
    } else {
        float p0, p1, p2;
        asm("s_nop 1\n\tv_mul_f32_dpp %0, %3, %4" LFS_NB(13) "v_mul_f32_dpp %1, %3, %5" LFS_NB(14) "v_mul_f32_dpp %2, %3, %6" LFS_NB(15)
            : "=&v"(p0), "=&v"(p1), "=&v"(p2) : "v"(rec), "v"(vc[0]), "v"(vc[1]), "v"(vc[2]));
        return (p0 + p1) + p2;
    }
}
#undef LFS_NB
#endif

// ray_eval (raster.hip) with the record in the row: the same operations in the same order, operand for operand
template <int MODE>
LFS_DI void ray_eval_row(const float rec, const f3& ro, const f3& d, RayEval& e) {
    e.om = {0.f, 0.f, 0.f};
    f3 gro{row_field<3>(rec), row_field<7>(rec), row_field<11>(rec)};
    if (MODE == RAY_ROLLING) {
        e.om = {ro.x - gro.x, ro.y - gro.y, ro.z - gro.z};
        gro = row_mat3(rec, e.om);
    }
    const f3 q = row_mat3(rec, d);
    const float l = fma3(q.x, q.x, q.y, q.y, q.z, q.z);
    const float rl = l > 0.f ? fast_rcp(l) : 0.f;
    e.t = fma3(gro.x, q.x, gro.y, q.y, gro.z, q.z) * rl;
    e.w = {__builtin_fmaf(-e.t, q.x, gro.x), __builtin_fmaf(-e.t, q.y, gro.y), __builtin_fmaf(-e.t, q.z, gro.z)};
    e.vis = __builtin_amdgcn_exp2f(-0.72134752044448170f * fma3(e.w.x, e.w.x, e.w.y, e.w.y, e.w.z, e.w.z));
}

// ---------------------------------------------------------------------------
// quadrant lists: split every cell list into the four lists of its 4x4 quadrants (same conic test, the quadrant's box of rays)
//   quad_list + 4 * (wpt * start + wl * len) + q * len,  len = the tile's list length;  quad_count[cell * 4 + q]
// ---------------------------------------------------------------------------
template <bool UNIFORM_ORIGIN>
__global__ void __launch_bounds__(256) raster_quad_lists_kernel(
    const uint32_t C, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block, const uint32_t cull_enabled,
    const CamDev* __restrict__ cams, const CullRec* __restrict__ cull, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t n_isects,
    const int32_t* __restrict__ cell_count, const int2* __restrict__ cell_list,
    int32_t* __restrict__ quad_count, int2* __restrict__ quad_list) {
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wpt = cells_per_tile(tile_size, false);
    const size_t cell = size_t(cc.tile_global) * wpt + cc.wl;
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1) ? n_isects : offsets[cc.tile_global + 1];
    const int32_t len = end - start;
    const bool tile_masked = masks != nullptr && !masks[cc.tile_global];
    const int32_t cnt = (tile_masked || len <= 0) ? 0 : cell_count[cell];
    if (cnt <= 0) { // uniform
        if (lane < 4) quad_count[cell * 4 + lane] = 0;
        return;
    }
    const RowCtx rc = row_ctx(cc);
    const CamDev& cam = cams[cc.cid];
    const float big = 3.0e38f;
    f3 ro, rd;
    const bool ok = cam_pixel_ray(cam, f2{float(rc.j) + 0.5f, float(rc.i) + 0.5f}, ro, rd);
    const bool act = rc.i < H && rc.j < W && ok;
    bool behind = false;
    float tu = 0.f, tv = 0.f;
    if (UNIFORM_ORIGIN) {
        const f3 cd = mul_t(cam.Rinv, rd);
        const bool front = cd.z > 0.f;
        behind = act && !front;
        const float iz = front ? 1.f / cd.z : 0.f;
        tu = cd.x * iz; tv = cd.y * iz;
    }
    const uint64_t act_b = __ballot(act);
    const bool wave_can_cull = UNIFORM_ORIGIN && cull_enabled != 0 && __ballot(behind) == 0ull;
    // per quadrant (= row): the box of its active rays, a quarter pixel wider
    const float mu_ = 0.25f / cam.fx, mv_ = 0.25f / cam.fy;
    const float r_ulo = row_min(act ? tu : big) - mu_, r_uhi = row_max(act ? tu : -big) + mu_;
    const float r_vlo = row_min(act ? tv : big) - mv_, r_vhi = row_max(act ? tv : -big) + mv_;
    float ulo[4], uhi[4], vlo[4], vhi[4];
    bool live[4], can[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ulo[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_ulo), 16 * q));
        uhi[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_uhi), 16 * q));
        vlo[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_vlo), 16 * q));
        vhi[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_vhi), 16 * q));
        live[q] = ((act_b >> (16 * q)) & 0xffffull) != 0ull;
        can[q] = wave_can_cull && (uhi[q] - ulo[q] < 1e30f) && (vhi[q] - vlo[q] < 1e30f);
    }
    const bool need_recs = UNIFORM_ORIGIN && cull_enabled != 0;
    const int2* __restrict__ in = cell_list + (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(len));
    int2* __restrict__ out = quad_list + 4 * (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(len));
    int32_t cq[4] = {0, 0, 0, 0};
    for (int32_t base = 0; base < cnt; base += 64) {
        const int32_t idx = base + int32_t(lane);
        const bool valid = idx < cnt;
        const int2 e = valid ? in[idx] : make_int2(0, 0);
        ConicRec k = conic_never();
        if (need_recs && valid) {
            const CullRec cr = cull[e.x];
            k = ConicRec{cr.a.x, cr.a.y, cr.a.z, cr.a.w, cr.b.x, cr.b.y, cr.b.z, cr.b.w};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bool hit = valid && live[q];
            if (can[q]) hit = hit && !conic_culled(k, ulo[q], uhi[q], vlo[q], vhi[q]);
            const uint64_t m = __ballot(hit);
            if (hit) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                out[size_t(q) * size_t(len) + size_t(cq[q]) + rank] = e;
            }
            cq[q] += __popcll(m);
        }
    }
    if (lane < 4) quad_count[cell * 4 + lane] = lane == 0 ? cq[0] : lane == 1 ? cq[1] : lane == 2 ? cq[2] : cq[3];
}

// Variant (lfs_set_debug_flags bit 3 on top of bit 2): the quadrant lists straight from the tile list in ONE kernel - raster_cull_kernel's
// batching (the workgroup gathers 64 x waves entries and their culling records once and shares them through LDS), four quadrant tests per
// entry instead of one cell test, no cell lists and no second gather. Measured (round 2): 0.143 ms against 0.201 ms for the split form, both slower than
// the 0.068 ms of the plain cell lists.
template <bool UNIFORM_ORIGIN>
__global__ void __launch_bounds__(256) raster_cull_quads_kernel(
    const uint32_t C, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block, const uint32_t cull_enabled,
    const CamDev* __restrict__ cams, const CullRec* __restrict__ cull, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ ids, const int32_t n_isects,
    int32_t* __restrict__ quad_count, int2* __restrict__ quad_list) {
    __shared__ float4 s_a[2][256], s_b[2][256];
    __shared__ int32_t s_g[2][256];
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return; // (uniform per workgroup)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wpt = cells_per_tile(tile_size, false);
    const size_t cell = size_t(cc.tile_global) * wpt + cc.wl;
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1) ? n_isects : offsets[cc.tile_global + 1];
    const int32_t len = end - start;
    const bool tile_masked = masks != nullptr && !masks[cc.tile_global];
    if (tile_masked || len <= 0) { // uniform per workgroup
        if (lane < 4) quad_count[cell * 4 + lane] = 0;
        return;
    }
    const RowCtx rc = row_ctx(cc);
    const CamDev& cam = cams[cc.cid];
    const float big = 3.0e38f;
    f3 ro, rd;
    const bool ok = cam_pixel_ray(cam, f2{float(rc.j) + 0.5f, float(rc.i) + 0.5f}, ro, rd);
    const bool act = rc.i < H && rc.j < W && ok;
    bool behind = false;
    float tu = 0.f, tv = 0.f;
    if (UNIFORM_ORIGIN) {
        const f3 cd = mul_t(cam.Rinv, rd);
        const bool front = cd.z > 0.f;
        behind = act && !front;
        const float iz = front ? 1.f / cd.z : 0.f;
        tu = cd.x * iz; tv = cd.y * iz;
    }
    const uint64_t act_b = __ballot(act);
    const bool cell_live = act_b != 0ull;
    const bool wave_can_cull = UNIFORM_ORIGIN && cull_enabled != 0 && __ballot(behind) == 0ull;
    const float mu_ = 0.25f / cam.fx, mv_ = 0.25f / cam.fy;
    const float r_ulo = row_min(act ? tu : big) - mu_, r_uhi = row_max(act ? tu : -big) + mu_;
    const float r_vlo = row_min(act ? tv : big) - mv_, r_vhi = row_max(act ? tv : -big) + mv_;
    float ulo[4], uhi[4], vlo[4], vhi[4];
    bool live[4], can[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ulo[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_ulo), 16 * q));
        uhi[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_uhi), 16 * q));
        vlo[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_vlo), 16 * q));
        vhi[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r_vhi), 16 * q));
        live[q] = ((act_b >> (16 * q)) & 0xffffull) != 0ull;
        can[q] = wave_can_cull && (uhi[q] - ulo[q] < 1e30f) && (vhi[q] - vlo[q] < 1e30f);
    }
    const bool need_recs = UNIFORM_ORIGIN && cull_enabled != 0; // workgroup-uniform
    int2* __restrict__ out = quad_list + 4 * (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(len));
    int32_t cq[4] = {0, 0, 0, 0};
    const int32_t E = int32_t(blockDim.x), nsub = E >> 6;
    auto fetch = [&](int32_t base, int32_t& g, CullRec& cr) {
        const int32_t i = base + int32_t(threadIdx.x);
        g = i < end ? ids[i] : 0;
        if (need_recs) cr = cull[g];
    };
    int32_t g_reg; CullRec cr_reg;
    cr_reg.a = make_float4(0.f, 0.f, 0.f, 0.f); cr_reg.b = cr_reg.a;
    fetch(start, g_reg, cr_reg);
    int buf = 0;
    for (int32_t base = start; base < end; base += E, buf ^= 1) {
        s_g[buf][threadIdx.x] = g_reg;
        if (need_recs) { s_a[buf][threadIdx.x] = cr_reg.a; s_b[buf][threadIdx.x] = cr_reg.b; }
        __syncthreads();
        if (base + E < end) fetch(base + E, g_reg, cr_reg);
        if (!cell_live) continue;
        for (int32_t sub = 0; sub < nsub; ++sub) {
            const int32_t slot = (sub << 6) + int32_t(lane);
            const int32_t my_idx = base + slot;
            if (base + (sub << 6) >= end) break; // uniform
            const bool valid = my_idx < end;
            const int2 e = make_int2(s_g[buf][slot], my_idx);
            ConicRec k = conic_never();
            if (need_recs) {
                const float4 a = s_a[buf][slot], b = s_b[buf][slot];
                k = ConicRec{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool hit = valid && live[q];
                if (can[q]) hit = hit && !conic_culled(k, ulo[q], uhi[q], vlo[q], vhi[q]);
                const uint64_t m = __ballot(hit);
                if (hit) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                    out[size_t(q) * size_t(len) + size_t(cq[q]) + rank] = e;
                }
                cq[q] += __popcll(m);
            }
        }
    }
    if (lane < 4) quad_count[cell * 4 + lane] = lane == 0 ? cq[0] : lane == 1 ? cq[1] : lane == 2 ? cq[2] : cq[3];
}

// The row's list walker state: entry k of the row's list (clamped to the last one; rows without entries read a dummy), the record dword
// of this lane. Vector loads return in order, so plain program order gives the compiler counted vmcnt waits: entries three and records
// two evaluations ahead.
struct RowList {
    const int2* __restrict__ ql; int32_t cnt; const float* __restrict__ recs; uint32_t l;
    LFS_DI int2 ent(int32_t pos) const { return cnt > 0 ? ql[pos] : make_int2(0, 0); }
    LFS_DI float rec(int32_t g) const { return recs[size_t(uint32_t(g)) * 16u + l]; }
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int CDIM, int MODE>
__global__ void __launch_bounds__(256) raster_fwd_rows_kernel(
    const uint32_t C, const uint32_t N, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block,
    const CamDev* __restrict__ cams, const GaussRec* __restrict__ recs, const float* __restrict__ colors,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ quad_count, const int2* __restrict__ quad_list, const int32_t n_isects,
    float* __restrict__ render_colors, float* __restrict__ render_alphas, int32_t* __restrict__ last_ids) {
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return;
    const RowCtx rc = row_ctx(cc);
    const uint32_t cid = cc.cid;
    const bool inside = rc.i < H && rc.j < W;
    const size_t pix_id = (size_t(cid) * H + rc.i) * W + rc.j;
    const float* bg = backgrounds ? backgrounds + cid * CDIM : nullptr;

    if (masks != nullptr && !masks[cc.tile_global]) { // as raster_fwd_kernel
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
    const bool ray_ok = lane_ray<MODE>(cam, rc.j, rc.i, ro, rd);
    const float INF = __builtin_inff();
    float thr = (inside && ray_ok) ? (1.f / 255.f) : INF; // see raster_fwd_kernel

    const uint32_t wpt = cells_per_tile(tile_size, false);
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1) ? n_isects : offsets[cc.tile_global + 1];
    const int32_t len = end - start;
    RowList rl;
    rl.ql = quad_list + 4 * (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(len)) + size_t(rc.row) * size_t(len);
    rl.cnt = quad_count[(size_t(cc.tile_global) * wpt + cc.wl) * 4 + rc.row];
    rl.recs = reinterpret_cast<const float*>(recs);
    rl.l = rc.l;
    const int32_t last = max(rl.cnt - 1, 0);
    const int32_t nmax = wave_max_i32(rl.cnt);

    float T = 1.f;
    float pix[CDIM];
#pragma unroll
    for (int k = 0; k < CDIM; ++k) pix[k] = 0.f;
    int32_t cur_idx = 0;

    int2 e0 = rl.ent(0), e1 = rl.ent(min(1, last)), e2 = rl.ent(min(2, last));
    float r0 = rl.rec(e0.x), r1 = rl.rec(e1.x);
    for (int32_t k = 0; k < nmax; ++k) {
        const bool mine = k < rl.cnt; // this row still has an entry at position k
        if ((k & 1) == 0 && __ballot(mine && thr < INF) == 0ull) break;
        const int2 e = e0;
        const float rec = r0;
        e0 = e1; e1 = e2; e2 = rl.ent(min(k + 3, last));
        r0 = r1; r1 = rl.rec(e1.x);

        RayEval re;
        ray_eval_row<MODE>(rec, ro, rd, re);
        const float alpha = fminf(0.999f, mul_field<12>(rec, re.vis));
        const bool pass = mine && !(alpha < thr);
        LFS_EMUL_COUNT(0);
        if (__ballot(pass) == 0ull) continue;
        LFS_EMUL_COUNT(1);
        const float next_T = T * (1.f - alpha);
        const bool fin = pass && next_T <= 1e-4f;
        const bool contrib = pass && !fin;
        const float vis = alpha * T;
        // (the DPP operands need the whole row: compute unconditionally, select afterwards)
        if (CDIM <= 3) {
            row_color_fma<CDIM>(rec, contrib ? vis : 0.f, pix); // (fma with 0 leaves pix[c] bit for bit: colours are finite)
        } else if (contrib) {
            const float* cp = colors + size_t(e.x) * CDIM;
#pragma unroll
            for (int c = 0; c < CDIM; ++c) pix[c] = __builtin_fmaf(cp[c], vis, pix[c]);
        }
        cur_idx = contrib ? e.y : cur_idx;
        T = contrib ? next_T : T;
        thr = fin ? INF : thr;
    }

    if (inside) {
        render_alphas[pix_id] = 1.f - T;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) render_colors[pix_id * CDIM + k] = bg ? pix[k] + T * bg[k] : pix[k];
        last_ids[pix_id] = cur_idx;
    }
}

// Sum 16 per-lane values over the 16 lanes of every row: lane l of a row ends with the row's total of v[l]. Each step pairs lanes that
// differ in one bit, keeps the half of the values that bit selects and adds the partner's copy of them: lane^8 (row_ror:8), the half-row
// mirror (flips bit 2), lane^2, lane^1: 8 + 4 + 2 + 1 = 15 additions, 45 VALU with the selects.
LFS_DI float row_transpose_sum16(const float (&v)[16], const uint32_t l) {
    const bool b3 = l & 8, b2 = l & 4, b1 = l & 2, b0 = l & 1;
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (b3 ? v[j + 8] : v[j]) + dpp_mov<DPP_ROR8>(b3 ? v[j] : v[j + 8]);
    float u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = (b2 ? w[j + 4] : w[j]) + dpp_mov<DPP_HALF_MIRROR>(b2 ? w[j] : w[j + 4]);
    const float s0 = (b1 ? u[2] : u[0]) + dpp_mov<DPP_XOR2>(b1 ? u[0] : u[2]);
    const float s1 = (b1 ? u[3] : u[1]) + dpp_mov<DPP_XOR2>(b1 ? u[1] : u[3]);
    return (b0 ? s1 : s0) + dpp_mov<DPP_XOR1>(b0 ? s0 : s1);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
template <int CDIM, int MODE, bool LOSS = false>
__global__ void __launch_bounds__(256) raster_bwd_rows_kernel(
    const uint32_t C, const uint32_t N, const uint32_t tw, const uint32_t th, const uint32_t W, const uint32_t H,
    const uint32_t tile_size, const uint32_t blocks_per_tile, const uint32_t waves_per_block,
    const CamDev* __restrict__ cams, const GaussRec* __restrict__ recs, const float* __restrict__ colors,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ masks,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ quad_count, const int2* __restrict__ quad_list, const int32_t n_isects,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ acc, float* __restrict__ v_colors_extra, const MseFuse mse = MseFuse{}) {
    const uint32_t n_tiles = tw * th, total_tiles = C * n_tiles;
    const CellCtx cc = cell_ctx(n_tiles, total_tiles, tw, tile_size, blocks_per_tile, waves_per_block);
    if (!cc.in_grid) return;
    const uint32_t cid = cc.cid;
    if (masks != nullptr && !masks[cc.tile_global]) return;
    const RowCtx rc = row_ctx(cc);
    const uint32_t lane = threadIdx.x & 63;
    const bool inside = rc.i < H && rc.j < W;
    const size_t pix_id = (size_t(cid) * H + rc.i) * W + rc.j;
    const float* bg = backgrounds ? backgrounds + cid * CDIM : nullptr;

    const CamDev& cam = cams[cid];
    f3 ro, rd;
    const bool ray_ok = lane_ray<MODE>(cam, rc.j, rc.i, ro, rd);
    const bool active = inside && ray_ok;

    const uint32_t wpt = cells_per_tile(tile_size, false);
    const int32_t start = offsets[cc.tile_global];
    const int32_t end = (cc.tile_global == total_tiles - 1) ? n_isects : offsets[cc.tile_global + 1];
    const int32_t len = end - start;
    RowList rl;
    rl.ql = quad_list + 4 * (size_t(wpt) * size_t(start) + size_t(cc.wl) * size_t(len)) + size_t(rc.row) * size_t(len);
    rl.cnt = quad_count[(size_t(cc.tile_global) * wpt + cc.wl) * 4 + rc.row];
    rl.recs = reinterpret_cast<const float*>(recs);
    rl.l = rc.l;

    float T_final = 1.f, v_ra = 0.f;
    int32_t bin_final = -1; // see raster_bwd_kernel
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
        if (lane == 0 && lsum != 0.f) unsafeAtomicAdd(mse.loss + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (LOSS_SLOTS - 1)), lsum * mse.scale);
    }
    float T = T_final;
    float tail = v_ra;
    if (bg) {
        float bd = 0.f;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) bd += bg[k] * vc[k];
        tail -= bd;
    }
    tail *= T_final;

    // per row: nothing behind the last contributor of its 16 pixels can matter: the number of list entries with index <= that
    const int32_t rmax = row_max_i32(bin_final);
    int32_t lo = 0, hi = rl.cnt;
    while (__ballot(lo < hi) != 0ull) { // (per-lane binary search, identical inside a row)
        const bool go = lo < hi;
        const int32_t mid = (lo + hi) >> 1;
        const int32_t y = go ? rl.ql[mid].y : 0;
        if (go) { if (y <= rmax) lo = mid + 1; else hi = mid; }
    }
    const int32_t n_walk = lo;
    const int32_t nmax = wave_max_i32(n_walk);
    if (nmax <= 0) return;
    const int32_t first = max(n_walk - 1, 0); // walk positions first, first - 1, ..., 0
    RowList rw = rl;
    rw.cnt = n_walk;

    int2 e0 = rw.ent(first), e1 = rw.ent(max(first - 1, 0)), e2 = rw.ent(max(first - 2, 0));
    float r0 = rw.rec(e0.x), r1 = rw.rec(e1.x);
    for (int32_t k = 0; k < nmax; ++k) {
        const bool mine = k < n_walk;
        const int2 e = e0;
        const float rec = r0;
        e0 = e1; e1 = e2; e2 = rw.ent(max(first - (k + 3), 0));
        r0 = r1; r1 = rw.rec(e1.x);

        RayEval re;
        ray_eval_row<MODE>(rec, ro, rd, re);
        const float vis = re.vis;
        const float opac = row_field<12>(rec);
        const float araw = opac * vis;
        const float alpha = fminf(0.999f, araw);
        const bool valid = mine && e.y <= bin_final && !(alpha < (1.f / 255.f));
        const uint64_t vb = __ballot(valid);
        LFS_EMUL_COUNT(2);
        if (vb == 0ull) continue;
        LFS_EMUL_COUNT(3);

        const float ra = fast_rcp(1.f - alpha);
        const float Tn = T * ra;
        T = valid ? Tn : T;
        const float fac = valid ? alpha * Tn : 0.f;
        float v[16], v_extra = 0.f, cv;
        if (CDIM <= 3) {
            cv = row_color_dot<CDIM>(rec, vc);
        } else {
            const float* cp = colors + size_t(e.x) * CDIM;
            cv = cp[0] * vc[0];
#pragma unroll
            for (int c = 1; c < CDIM; ++c) cv = __builtin_fmaf(cp[c], vc[c], cv);
        }
        const float v_alpha = __builtin_fmaf(ra, tail - Bsum, Tn * cv);
        Bsum = __builtin_fmaf(fac, cv, Bsum);
#pragma unroll
        for (int c = 0; c < CDIM; ++c) {
            const float vrgb = fac * vc[c];
            if (c < 3) v[13 + c] = vrgb; else v_extra = vrgb;
        }
#pragma unroll
        for (int c = CDIM; c < 3; ++c) v[13 + c] = 0.f;
        const float v_op = (valid && araw <= 0.999f) ? vis * v_alpha : 0.f;
        v[12] = v_op;
        const float sgeo = opac * v_op;
        const f3 a = re.w * sgeo;
        const f3 vg = a * re.t;
        v[0] = vg.x * rd.x; v[1] = vg.x * rd.y; v[2] = vg.x * rd.z;
        v[3] = vg.y * rd.x; v[4] = vg.y * rd.y; v[5] = vg.y * rd.z;
        v[6] = vg.z * rd.x; v[7] = vg.z * rd.y; v[8] = vg.z * rd.z;
        if (MODE == RAY_ROLLING) {
            const f3& om = re.om;
            v[0] -= a.x * om.x; v[1] -= a.x * om.y; v[2] -= a.x * om.z;
            v[3] -= a.y * om.x; v[4] -= a.y * om.y; v[5] -= a.y * om.z;
            v[6] -= a.z * om.x; v[7] -= a.z * om.y; v[8] -= a.z * om.z;
        }
        v[9] = a.x; v[10] = a.y; v[11] = a.z;
        const float total = row_transpose_sum16(v, rc.l);        // value l of this row's Gaussian
        const bool row_has = row_bits(vb, rc.row) != 0u;         // rows without a valid pixel add nothing
        if (row_has) unsafeAtomicAdd(acc + size_t(uint32_t(e.x)) * ACC_STRIDE + rc.l, total);
        if (CDIM > 3) {
            const float ex = row_sum(v_extra);
            if (row_has && rc.l == 0) unsafeAtomicAdd(v_colors_extra + size_t(uint32_t(e.x)) * CDIM + 3, ex);
        }
    }
}

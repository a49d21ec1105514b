// Device-side building blocks shared by the two rasterizers (raster.hip: world-space 3DGUT; fastgs.hip: EWA splatting):
// the 64-byte wave-uniform record, the workgroup -> (tile, 8x8 cell) mapping, the scalar-unit list walker and the
// 16-value wave reduction. See raster.hip for the design notes.
#pragma once
#include "lfs_math.cuh"

namespace lfs {

struct __attribute__((aligned(16))) GaussRec { float4 r0, r1, r2, r3; };

constexpr int ACC_STRIDE = 16; // LFS_ACC_SYM (global shutter): B'' (6) | a'' (3) | - - - | opacity | rgb(3); otherwise A(9) | G(3) | opacity | rgb(3)

// LFS_REC_LOG2 (round 3, the 3DGUT rasterizer): records carry M' = c M, g' = c g with c = sqrt(0.5 log2 e), and log2(opacity) in place of the opacity, so that
//   alpha_raw = opac * exp(-0.5 |w|^2) = exp2(log2(opac) - |w'|^2),   w' = g' - t q' = c w  (t is scale-free)
// costs the evaluation 3 fma + 1 exp instead of 3 fma + 2 mul + 1 exp, and the backward works on s = alpha_raw * dL/dalpha directly (it never needs vis or
// the opacity on their own). What the backward accumulates is then  A' = c A,  G' = c G  and  O' = opac * dL/dopac: the finish kernels undo the two
// per-Gaussian constants when they read the row (REC_UNSCALE, 1 / opac). 0 = the round-1/2 records (un-scaled, plain opacity).
#ifndef LFS_REC_LOG2
#define LFS_REC_LOG2 1
#endif
// LFS_REC_ROT (round 6, global shutter only): the record is stored in a ROTATED Gaussian frame U (rot_frame below) in which g = M (o - mu) points along the third axis,
// g'' = U g = (0, 0, G). Rotations leave the distance of the ray q = M d to the origin untouched, and with two components of g'' gone it collapses:
//   t = G q''.z / |q''|^2,   w'' = g'' - t q'' = (-t q''.x, -t q''.y, G m / l),   |w''|^2 = G^2 m / l,   m = q''.x^2 + q''.y^2,  l = m + q''.z^2
// (|g| sin(angle between the ray and the direction to the centre)). The forward needs neither t nor w: 9 + 3 + rcp + 3 + exp2 instructions instead of 9 + 3 + rcp + 3 + 1 + 3 + 3
// + exp2; the backward gets a foot vector with NO difference of large numbers in it - the Gram-Schmidt step of LFS_BWD_REORTH (7 instructions) has nothing left to remove -
// and accumulates dL/dA'' = U dL/dA, dL/dg'' = U dL/dg exactly as before; finish_geometry (raster.hip) multiplies the two sums by U^T once per Gaussian.
// Record: rows of U M' (M' = c M Rinv) in r0..r2.xyz, r0.w = G^2, r1.w = 0, r2.w = G (G = c |g|). 0 = the round-3 .. 6 records (g in the .w fields).
#ifndef LFS_REC_ROT
#define LFS_REC_ROT 1
#endif
// LFS_REC_PKQ (with LFS_REC_ROT, global shutter): rows 0 / 1 of the record's matrix interleaved by column, q.x and q.y as ONE chain of v_pk_mul_f32 + 2 v_pk_fma_f32 with
// the column pairs as SGPR-pair operands - the same products and roundings per component (bit-identical images and gradients), 3 instructions instead of 6.
#ifndef LFS_REC_PKQ
#define LFS_REC_PKQ 0
#endif
#if LFS_REC_PKQ && !LFS_REC_ROT
#error "LFS_REC_PKQ is a layout of the rotated records"
#endif
#if LFS_REC_ROT && !LFS_REC_LOG2
#error "LFS_REC_ROT is written for the LFS_REC_LOG2 records"
#endif
// LFS_ACC_SYM (round 6, with LFS_REC_ROT): what the backward accumulates per Gaussian. With a = s w (s = alpha_raw dL/dalpha_raw) the gradients of the record are
// dL/dq = t a and dL/dg = -a, so dL/dM = sum (t a (x) d - a (x) (o - mu)) = -sum a (x) (o - mu - t d) - and o - mu - t d is M^-1 w, the closest point of the ray in
// world space. Hence dL/dM = -B M^-T with the SYMMETRIC B = sum s w w^T: six sums instead of the nine of dL/dA plus three of them again in dL/dg, no ray direction and
// no second use of t in the products (10 VALU instead of 15 per evaluation, 13 LDS rows instead of 16), and dL/dscale_c = B_cc / s_c comes out of the finish pass WITHOUT
// the difference it used to be: for a flat Gaussian the thin axis' dL/dA . M and dL/dg . g are 1e6 times their sum, which is why dL/dscales of the thin axis was
// "fp32-limited for every implementation" (the reference's per-pixel form included: tools/aniso_probe.py, 3e-4 .. 2e-3 at aspect 80). The finish pass rotates B'' and
// sum a'' back with a single-precision frame: a frame 1e-7 away from the record's now mixes 1e-7 of B's large entries into small ones that nothing amplifies.
// Row: B''xx, B''yy, B''xz, B''yz, B''xy, B''zz | a''x, a''y, a''z | - - - | opac dL/dopac | dL/drgb. 0 = dL/dA (9) | -dL/dg (3) | ... (rounds 1 - 6; the rolling-shutter kernels keep it).
#ifndef LFS_ACC_SYM
#define LFS_ACC_SYM LFS_REC_ROT
#endif
#if LFS_ACC_SYM && !LFS_REC_ROT
#error "LFS_ACC_SYM needs the rotated records (LFS_REC_ROT)"
#endif
constexpr float REC_SCALE = 0.84932180028801904f;   // sqrt(0.5 * log2(e))
constexpr float REC_UNSCALE = 1.17741002251547469f; // 1 / REC_SCALE = sqrt(2 ln 2)


// ---------------------------------------------------------------------------
// tile / cell bookkeeping shared by cull, fwd and bwd
// ---------------------------------------------------------------------------
struct CellCtx {
    uint32_t cid, tile_global, wl, i, j; // wl = 8x8 cell index inside the tile (wave-uniform)
    bool in_grid;
};
// Workgroup -> (tile, cell). Consecutive workgroup ids go round-robin over the 8 XCDs, and a workgroup cannot move to another XCD: remap so that each XCD
// works on contiguous chunks of tiles (the records of neighbouring tiles meet in the same L2) AND every XCD gets the same share of the image's work.
// One chunk per XCD (rounds 1-2) gave the XCDs with the top and bottom eighth of a view half the work of the others - on SYN-B the busiest XCD had 1.16 - 1.17 x
// the mean (tools/band_balance.py), so the kernels ran 16 % longer than their work. LFS_XCD_BANDS chunks per XCD, dealt round-robin (chunk c -> XCD c % 8):
// 8 of them (a chunk = about one tile row of a 1080p view) bring that to 1.01 - 1.02.
#ifndef LFS_XCD_BANDS
#define LFS_XCD_BANDS 8
#endif
// workgroups per chunk: whole tiles
__host__ __device__ inline uint32_t cell_chunk_blocks(uint32_t nb, uint32_t blocks_per_tile) {
    const uint32_t chunks = 8u * LFS_XCD_BANDS;
    const uint32_t cs = (nb + chunks - 1) / chunks;
    return cs ? ((cs + blocks_per_tile - 1) / blocks_per_tile) * blocks_per_tile : blocks_per_tile;
}
// the grid that covers nb workgroups under this mapping (what the host launches)
__host__ __device__ inline uint32_t cell_grid_blocks(uint64_t nb, uint32_t blocks_per_tile) {
    return 8u * LFS_XCD_BANDS * cell_chunk_blocks(uint32_t(nb), blocks_per_tile);
}
LFS_DI CellCtx cell_ctx(uint32_t n_tiles, uint32_t total_tiles, uint32_t tw, uint32_t tile_size, uint32_t blocks_per_tile, uint32_t waves_per_block) {
    CellCtx c;
    const uint32_t nb = total_tiles * blocks_per_tile;
    const uint32_t cs = cell_chunk_blocks(nb, blocks_per_tile);
    const uint32_t k = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const uint32_t band = k / cs;
    const uint32_t b = (band * 8u + xcd) * cs + (k - band * cs);
    c.in_grid = b < nb && band < LFS_XCD_BANDS;
    const uint32_t tg = b / blocks_per_tile, bt = b % blocks_per_tile;
    c.tile_global = tg;
    c.cid = tg / n_tiles;
    const uint32_t tile = tg % n_tiles;
    const uint32_t ty = tile / tw, tx = tile % tw;
    const uint32_t wps = tile_size >> 3; // 8x8 cells per tile side
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.wl = bt * waves_per_block + wave;
    const uint32_t lane = threadIdx.x & 63;
    c.i = ty * tile_size + (c.wl / wps) * 8 + (lane >> 3);
    c.j = tx * tile_size + (c.wl % wps) * 8 + (lane & 7);
    return c;
}

LFS_DI float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
LFS_DI float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// a * b with 0 * anything (inf, NaN) = 0 (v_mul_legacy_f32): the reciprocal of a zero length needs no clamp before it meets the zero numerator. Through the LLVM
// intrinsic, NOT through inline asm: the compiler has to see the instruction to keep the wait state gfx950 needs between a transcendental (v_rcp_f32) and a VALU
// instruction that reads its result - the first form of this (asm) was scheduled right behind the v_rcp_f32 in raster_fwd_kernel and read the stale register: images
// off by 5e-2, on the hardware only (profiles/r06/lease21_rot_frame_ab.txt).
#ifndef LFS_EMULATE
extern "C" __device__ float lfs_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
#endif
LFS_DI float mul_zero(float a, float b) {
#ifdef LFS_EMULATE
    return (a == 0.f || b == 0.f) ? 0.f : a * b;
#else
    return lfs_fmul_legacy(a, b);
#endif
}

// LFS_SEL_E64 (round 6): the per-lane conditions of the inner loops as LANE MASKS in SGPR pairs - v_cmp_*_e64 writes the mask, plain SALU combines masks, the
// wave-level "nobody" test is s_cmp on the mask, and every select is v_cndmask_b32_e64 on a mask - instead of the compiler's VCC / EXEC forms. tools/valu_rate.hip on
// the MI355X (profiles/r06/valu_rate.json), ns per wavefront and SIMD at 7 wavefronts per SIMD: v_cndmask_b32_e64 with an SGPR-pair mask 1.8 (the plain VALU rate);
// v_cmp -> VCC -> v_cndmask_b32_e32 6.1 per pair (3.7 would be the sum of its parts); v_cmp + s_cbranch_vccz 5.1; s_and_saveexec_b64 + s_or_b64 exec 7.3 per pair.
// Same compares, same selects, same bits. The helpers take the condition twice: as a bool (the emulator build and the switch-off form) and as the mask.
#ifndef LFS_SEL_E64
#define LFS_SEL_E64 0
#endif
typedef unsigned long long lmask_t;
#if LFS_SEL_E64 && !defined(LFS_EMULATE)
#define LFS_MASK_ASM 1
#else
#define LFS_MASK_ASM 0
#endif
LFS_DI lmask_t lane_mask(const bool c) {
#ifdef LFS_EMULATE
    return __ballot(c);
#else
    return __builtin_amdgcn_ballot_w64(c);
#endif
}
// a < b is FALSE (NaN passes), a <= b, uniform int k <= per-lane int v
LFS_DI lmask_t mask_nlt_f32(const float a, const float b) {
#if LFS_MASK_ASM
    lmask_t m; asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m;
#else
    return lane_mask(!(a < b));
#endif
}
LFS_DI lmask_t mask_le_f32(const float a, const float b) {
#if LFS_MASK_ASM
    lmask_t m; asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m;
#else
    return lane_mask(a <= b);
#endif
}
LFS_DI lmask_t mask_le_i32_uniform(const int32_t k_uniform, const int32_t v) {
#if LFS_MASK_ASM
    lmask_t m; asm("v_cmp_le_i32_e64 %0, %1, %2" : "=s"(m) : "s"(k_uniform), "v"(v)); return m;
#else
    return lane_mask(k_uniform <= v);
#endif
}
LFS_DI float sel_mask(const lmask_t m, const float if_set, const float if_clear) {
#if LFS_MASK_ASM
    float r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m)); return r;
#else
    return ((m >> (threadIdx.x & 63u)) & 1ull) ? if_set : if_clear;
#endif
}
LFS_DI int32_t sel_mask_i32(const lmask_t m, const int32_t if_set, const int32_t if_clear) {
#if LFS_MASK_ASM
    int32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m)); return r;
#else
    return ((m >> (threadIdx.x & 63u)) & 1ull) ? if_set : if_clear;
#endif
}
LFS_DI float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
LFS_DI float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
LFS_DI float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// One Gaussian against one ray, shared by fwd and bwd so that both see bit-identical alphas. Every sum of
// products is an explicit fma chain: with -ffp-contract=fast alone the compiler is free to pick WHICH product
// of a*b + c*d it fuses, and it picks differently in different inlined copies - results would depend on the
// position of a Gaussian in the list (measured: 1 ulp), and the culling on/off bit-identity test would fail.
LFS_DI float fma3(float ax, float bx, float ay, float by, float az, float bz) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}
LFS_DI f3 cross_fma(const f3& a, const f3& b) {
    return {__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)), __builtin_fmaf(a.x, b.y, -(b.x * a.y))};
}
// The frame of LFS_REC_ROT: an orthonormal U (rows) whose third axis is g / |g|, U g = (0, 0, |g|); g = 0 or not finite: some frame, nothing depends on which. Duff et al.
// 2017 ("Building an orthonormal basis, revisited"): branch-free, no cancellation for any direction. The RECORD's frame is built in DOUBLE precision, with the product
// U M' it is used in: in single precision U g is (0, 0, |g|) only to 6e-8 |g|, and for a flat Gaussian |g| is 1e4 where the foot vector is 1 - an offset of 6e-4 in w
// that the evaluation knows nothing about (measured: tools/aniso_probe.py, dL/dmeans 5e-3 off at aspect 80, the forward 2.6e-4; profiles/r06/lease21). Single-precision
// v_rsq / v_rcp seeds with two Newton steps in double instead of IEEE sqrt / division. What else was measured on the way (lease 21 - 23, with the dL/dA accumulators of
// LFS_ACC_SYM = 0, whose finish pass needs the record's frame bit for bit): the double frame rebuilt in the fused tail kernel took it from 224 to 258 VGPRs and ONE wavefront
// per SIMD (0.29 -> 0.40 ms); a single-precision frame there doubled the error of dL/dscales (1e-7 of dL/dA's large rows lands in the thin one); one single-precision frame
// on both sides with its residual sheared away strains t against the |g| = 1e4 it multiplies (dL/dscales 2e-3 off at aspect 80).
struct RotFrame { double nx, ny, nz, b, c00, c11, sgb, sgnx, len; };
LFS_DI void rot_frame(const f3 g, RotFrame& F) {
#pragma clang fp contract(off)
    const double gx = g.x, gy = g.y, gz = g.z;
    const double len2 = gx * gx + gy * gy + gz * gz;
    double nx = 0.0, ny = 0.0, nz = 1.0;
    F.len = 0.0;
    if (len2 > 1e-36 && len2 < 1e36) {   // (|g| within 1e-18 .. 1e18: inside the range of the single-precision seed)
        double inv = double(fast_rsq(float(len2)));
        inv = inv * (1.5 - 0.5 * len2 * inv * inv);
        inv = inv * (1.5 - 0.5 * len2 * inv * inv);
        F.len = len2 * inv; nx = gx * inv; ny = gy * inv; nz = gz * inv;
    }
    const double sg = nz >= 0.0 ? 1.0 : -1.0;
    const double den = sg + nz;            // |den| in [1, 2]
    double r = double(fast_rcp(float(den)));
    r = r * (2.0 - den * r);
    r = r * (2.0 - den * r);
    const double a = -r;
    F.nx = nx; F.ny = ny; F.nz = nz;
    F.b = nx * ny * a; F.c00 = 1.0 + sg * nx * nx * a; F.c11 = sg + ny * ny * a; F.sgb = sg * F.b; F.sgnx = sg * nx;
}
// rows of U: (c00, sgb, -sgnx), (b, c11, -ny), (nx, ny, nz)
LFS_DI void rot_apply(const RotFrame& F, const float v0, const float v1, const float v2, float& o0, float& o1, float& o2) {   // U v
#pragma clang fp contract(off)
    const double x = v0, y = v1, z = v2;
    o0 = float(F.c00 * x + F.sgb * y - F.sgnx * z);
    o1 = float(F.b * x + F.c11 * y - F.ny * z);
    o2 = float(F.nx * x + F.ny * y + F.nz * z);
}
LFS_DI void rot_apply_t(const RotFrame& F, const float v0, const float v1, const float v2, float& o0, float& o1, float& o2) { // U^T v (the finish pass of LFS_ACC_SYM = 0: one vector at a time between scheduling barriers, for the tail kernel's register budget)
#pragma clang fp contract(off)
    const double x = v0, y = v1, z = v2;
    o0 = float(F.c00 * x + F.b * y + F.nx * z);
    o1 = float(F.sgb * x + F.c11 * y + F.ny * z);
    o2 = float(F.nz * z - F.sgnx * x - F.ny * y);
#ifndef LFS_EMULATE
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// The same frame in single precision: the finish pass of LFS_ACC_SYM (see there for why 1e-7 of disagreement with the record's frame is harmless on that side)
LFS_DI void rot_frame_f32(const f3 g, m3& U) {
#pragma clang fp contract(off)
    const float len2 = g.x * g.x + g.y * g.y + g.z * g.z;
    float nx = 0.f, ny = 0.f, nz = 1.f;
    if (len2 > 1e-36f && len2 < 1e36f) { const float inv = fast_rsq(len2); nx = g.x * inv; ny = g.y * inv; nz = g.z * inv; }
    const float sg = nz >= 0.f ? 1.f : -1.f;
    const float a = -fast_rcp(sg + nz), b = nx * ny * a;
    U.m[0][0] = 1.f + sg * nx * nx * a; U.m[0][1] = sg * b; U.m[0][2] = -sg * nx;
    U.m[1][0] = b; U.m[1][1] = sg + ny * ny * a; U.m[1][2] = -ny;
    U.m[2][0] = nx; U.m[2][1] = ny; U.m[2][2] = nz;
}

// Walk a cell list with the records arriving through the SCALAR unit: two groups of two record
// buffers in SGPRs. Scalar loads return out of order, so every wait is s_waitcnt lgkmcnt(0): the loop
// waits for group B right BEFORE refilling group A (and vice versa), which gives each record load two full
// evaluations (~300 cycles) in flight and never copies a buffer. Entries are visited at positions
// first, first+step, ... (n of them); eval(rec, entry) per entry; alive() is polled every two entries.
// (LFS_EMULATE: the host build of tests/emul - no registers to pin there)
#ifdef LFS_EMULATE
#define LFS_SGPR_PIN2(text, a, b) ((void)(a), (void)(b))
#else
#define LFS_SGPR_PIN2(text, a, b) asm volatile(text ::"s"(a), "s"(b))
#endif
// LFS_WALK_PREFETCH (round 6): the scalar record loads above are in flight for two evaluations; a record that misses the XCD's L2 (FETCH_SIZE says most of them do:
// raster_fwd moves 64 B from the fabric per walked entry) takes longer than that under load. With the switch on, the 64 lanes pull the records of the NEXT 64 entries
// towards the L2 with one vector load per chunk (lane = entry; the value is never used), so that the scalar load finds the line there. 0 = off.
#ifndef LFS_WALK_PREFETCH
#define LFS_WALK_PREFETCH 0
#endif
template <int STEP, class Eval, class Alive>
LFS_DI void walk_cell_list(const int2* __restrict__ cl, const GaussRec* __restrict__ recs, const int32_t first, const int32_t n,
                           Eval&& eval, Alive&& alive) {
    if (n <= 0) return;
    const int32_t last = n - 1;
#if LFS_WALK_PREFETCH && !defined(LFS_EMULATE)
    const int32_t pf_lane = int32_t(threadIdx.x & 63u);
    auto pf_id = [&](int32_t k) { return k < n ? reinterpret_cast<const int2*>(reinterpret_cast<const char*>(cl) + (uint32_t(first + STEP * k) << 3))->x : -1; };
    auto pf_touch = [&](int32_t g) { return g >= 0 ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(recs) + (uint32_t(g) << 6)) : 0.f; };
    float pf_val = pf_touch(pf_id(pf_lane));   // entries 0 .. 63 (the first four also arrive through the scalar loads below)
    int32_t pf_next = pf_id(64 + pf_lane);     // ids of entries 64 .. 127: touched when the walk reaches entry 0 + 32
#endif
    // unsigned 32-bit BYTE offsets: the scalar load then takes (64-bit base, 32-bit offset register) and the per-entry address
    // arithmetic is one shift instead of a sign extension + 64-bit shift + 64-bit add (the EWA forward, 22 VALU per entry, was
    // limited by its ~19 SALU per entry). Limits: C*N < 2^26 records, cell list < 2^29 entries (checked by the callers).
    auto ent = [&](int32_t k) { return *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(cl) + (uint32_t(first + STEP * min(k, last)) << 3)); };
    auto rec_at = [&](int32_t g) { return *reinterpret_cast<const GaussRec*>(reinterpret_cast<const char*>(recs) + (uint32_t(g) << 6)); };
    int2 eA0 = ent(0), eA1 = ent(1), eB0 = ent(2), eB1 = ent(3);
    int2 nA0 = ent(4), nA1 = ent(5), nB0 = make_int2(0, 0), nB1 = make_int2(0, 0);
    GaussRec A0 = rec_at(eA0.x), A1 = rec_at(eA1.x), B0 = rec_at(eB0.x), B1 = rec_at(eB1.x);
    for (int32_t k = 0; k < n; k += 4) {
        if (!alive()) break;
#if LFS_WALK_PREFETCH && !defined(LFS_EMULATE)
        if ((k & 63) == 32 && k + 32 < n) { // (uniform) half a chunk before the walk gets there: the next chunk's records, and the ids of the one after
            asm volatile("" ::"v"(pf_val));  // the previous touch has long landed; this only keeps its load alive
            pf_val = pf_touch(pf_next);
            pf_next = pf_id(k + 96 + pf_lane);
        }
#endif
        eval(A0, eA0);
        if (k + 1 < n) eval(A1, eA1);
        LFS_SGPR_PIN2("; group B must have landed before group A is refilled", B0.r0.x, B1.r0.x);
        eA0 = nA0; eA1 = nA1;
        A0 = rec_at(eA0.x); A1 = rec_at(eA1.x);
        nB0 = ent(k + 6); nB1 = ent(k + 7);
        if (k + 2 >= n || !alive()) break;
        eval(B0, eB0);
        if (k + 3 < n) eval(B1, eB1);
        LFS_SGPR_PIN2("; group A must have landed before group B is refilled", A0.r0.x, A1.r0.x);
        eB0 = nB0; eB1 = nB1;
        B0 = rec_at(eB0.x); B1 = rec_at(eB1.x);
        nA0 = ent(k + 8); nA1 = ent(k + 9);
    }
#if LFS_WALK_PREFETCH && !defined(LFS_EMULATE)
    asm volatile("" ::"v"(pf_val), "v"(pf_next)); // (the touches are loads the compiler would otherwise drop)
#endif
}

template <int CTRL>
LFS_DI float dpp_mov(float v) { // row-local lane permutation (DPP): 0xB1 = lane^1, 0x4E = lane^2, 0x124 / 0x128 = rotate by 4 / 8
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}

// Sum 16 per-lane values over the 64 lanes and add the 16 totals to dst[0..15] with one 16-lane atomic instruction.
// Every step but the last two HALVES the number of live values while folding lanes: v_permlane32_swap (lane halves),
// v_permlane16_swap (row pairs), then lane^1 and lane^2 inside the quads (select + DPP add); two row rotations finish.
// 35 VALU for 16 sums (a butterfly per value would be 16 x 6).
// ACC selects what happens to the 16 totals. 0 (default): float atomics - the order in which the wavefronts of different cells reach a
// Gaussian's row is not reproducible, so neither are the last bits of the sums (the reference's atomicAdd backward has the same property).
// 1 / 2: the two passes of the DETERMINISTIC mode (lfs_set_debug_flags bit 4; tests and the PSNR comparison of DESIGN.md): pass 1 takes the
// maximum of |total| per slot with an integer atomicMax on the float bits (order-independent), pass 2 converts every total to fixed point
// with 40 fractional bits below that maximum's exponent and adds it with a 64-bit integer atomic (exact, associative): two runs give
// bit-identical gradients. det64 = the int64 accumulator rows (16 per Gaussian), dst doubles as the uint32 maxima in passes 1 / 2.
template <int ACC = 0>
LFS_DI void wave_sum16_atomic(const float (&v)[16], float* __restrict__ dst, const uint32_t lane, unsigned long long* __restrict__ det64 = nullptr) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 8]), false, false);
        w[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    float u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[j]), __float_as_uint(w[j + 4]), false, false);
        u[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // row r (16 lanes) now holds partial sums of v[4r .. 4r+3] in u[0..3]
    const bool b0 = lane & 1, b1 = lane & 2;
    // lane^1: even lanes keep (u0, u1), odd lanes keep (u2, u3)
    const float s0 = (b0 ? u[2] : u[0]) + dpp_mov<0xB1>(b0 ? u[0] : u[2]);
    const float s1 = (b0 ? u[3] : u[1]) + dpp_mov<0xB1>(b0 ? u[1] : u[3]);
    // lane^2: bit1 == 0 keeps the first of the pair, bit1 == 1 the second
    float t = (b1 ? s1 : s0) + dpp_mov<0x4E>(b1 ? s0 : s1);
    t += dpp_mov<0x124>(t);
    t += dpp_mov<0x128>(t);
    // lane L holds the total of v[4 * (L >> 4) + 2 * (L & 1) + ((L >> 1) & 1)]
    if ((lane & 12) == 0) {
        const uint32_t slot = 4 * (lane >> 4) + 2 * (lane & 1) + ((lane >> 1) & 1);
        if (ACC == 0) unsafeAtomicAdd(dst + slot, t);
#ifndef LFS_EMULATE
        else if (ACC == 1) atomicMax(reinterpret_cast<uint32_t*>(dst) + slot, __float_as_uint(t) & 0x7fffffffu);
        else {
            const uint32_t mbits = reinterpret_cast<const uint32_t*>(dst)[slot];
            if (mbits != 0u && t != 0.f) {
                const int e = max(int((mbits >> 23) & 0xffu), 1) - 127;                 // exponent of the slot's largest |total|
                atomicAdd(det64 + slot, (unsigned long long)__float2ll_rn(ldexpf(t, 40 - e))); // |fixed| < 2^41; two's complement adds
            }
        }
#endif
    }
}


// The same reduction on PACKED pairs (v_pk_add_f32: two adds per instruction at ~1.15 issue slots): the caller keeps its 16 values as eight aligned register
// pairs V[j] = (v[2j], v[2j+1]); v_permlane32_swap / v_permlane16_swap exchange the components of pair j with those of pair j + 4 (resp. j + 2) in place, so
// the two halves to add are register pairs again. Same values, same order of additions per slot as wave_sum16_atomic: bit-identical totals.
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef LFS_BWD_LIMITER_EXPERIMENT
#define LFS_BWD_LIMITER_EXPERIMENT 0
#endif
template <int ACC = 0>
LFS_DI void wave_sum16_atomic_pk(const v2f (&V)[8], float* __restrict__ dst, const uint32_t lane, unsigned long long* __restrict__ det64 = nullptr) {
#if LFS_BWD_LIMITER_EXPERIMENT == 2   // (measurement only: no cross-lane reduction, no atomic)
    {
        v2f acc2 = V[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc2 += V[j];
        if (acc2.x + acc2.y == 123.456f) dst[lane & 15] = acc2.x;
        return;
    }
#endif
    v2f W[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { // v[2j], v[2j+1] with v[2j+8], v[2j+9]
        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(V[j].x), __float_as_uint(V[j + 4].x), false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(V[j].y), __float_as_uint(V[j + 4].y), false, false);
        W[j] = v2f{__uint_as_float(r0[0]), __uint_as_float(r1[0])} + v2f{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
    }
    // W[j] = (w[2j], w[2j+1]) of wave_sum16_atomic; its second level pairs w[j] with w[j+4]: pair j with pair j + 2
    v2f U[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(W[j].x), __float_as_uint(W[j + 2].x), false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(W[j].y), __float_as_uint(W[j + 2].y), false, false);
        U[j] = v2f{__uint_as_float(r0[0]), __uint_as_float(r1[0])} + v2f{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
    }
    const float u[4] = {U[0].x, U[0].y, U[1].x, U[1].y};
    const bool b0 = lane & 1, b1 = lane & 2;
    const float s0 = (b0 ? u[2] : u[0]) + dpp_mov<0xB1>(b0 ? u[0] : u[2]);
    const float s1 = (b0 ? u[3] : u[1]) + dpp_mov<0xB1>(b0 ? u[1] : u[3]);
    float t = (b1 ? s1 : s0) + dpp_mov<0x4E>(b1 ? s0 : s1);
    t += dpp_mov<0x124>(t);
    t += dpp_mov<0x128>(t);
    if ((lane & 12) == 0) {
        const uint32_t slot = 4 * (lane >> 4) + 2 * (lane & 1) + ((lane >> 1) & 1);
#if LFS_BWD_LIMITER_EXPERIMENT == 1   // (measurement only, wrong results: where does the time go? no atomic - a store nobody orders)
        if (t == 123.456f) dst[slot] = t;
#else
        if (ACC == 0) unsafeAtomicAdd(dst + slot, t);
#ifndef LFS_EMULATE
        else if (ACC == 1) atomicMax(reinterpret_cast<uint32_t*>(dst) + slot, __float_as_uint(t) & 0x7fffffffu);
        else {
            const uint32_t mbits = reinterpret_cast<const uint32_t*>(dst)[slot];
            if (mbits != 0u && t != 0.f) {
                const int e = max(int((mbits >> 23) & 0xffu), 1) - 127;
                atomicAdd(det64 + slot, (unsigned long long)__float2ll_rn(ldexpf(t, 40 - e)));
            }
        }
#endif
#endif
    }
}

// The same 16 sums through LDS (round 3; LFS_BWD_LDS_REDUCE): every lane parks its 16 values in its own row of a [64][17] scratch block of the wavefront
// (ds_write2_b32 x 8; the odd row stride spreads the lanes over all banks), lane L then sums column L & 15 over the 16 rows of its quarter L >> 4
// (ds_read2_b32 x 8 + a 4-level tree), and two swaps fold the four quarters. 12 VALU instructions + 16 LDS instructions instead of the ~45 issue slots of the
// register transpose above (12 half-rate swaps, 6 selects, 5 DPP adds and the wait states both need): the backward is VALU-issue bound, its LDS pipe idle.
// A wavefront's LDS operations execute in order, so the write -> read -> (next entry's) write sequence needs no barrier.
constexpr int RED_STRIDE = 17;
// LFS_RED_ADDTID (round 3, second form): the block is stored VALUE-major, [16][72] floats - value k of lane l at k * 72 + l - with ds_write_addtid_b32
// (address = M0 + offset + 4 * lane: no address VGPR, 2 LDS cycles per instruction instead of 6 for ds_write2_b32: MI355X_MICROARCH.md, LDS), and lane L
// (k = L & 15, q = L >> 4) reads the four 16-byte pieces {16 j + 4 q .. + 3}, j = 0..3, of row k with ds_read_b128 (row stride 72: the 16 lanes of every
// b128 service group hit 16 disjoint 4-bank ranges). 16 x 2 + 4 x 4 = 48 LDS cycles per evaluation instead of 8 x 6 + 8 x 4 = 80.
#ifndef LFS_RED_ADDTID
#define LFS_RED_ADDTID 1
#endif
constexpr int RED_ROW = 72;
constexpr int RED_SCRATCH_FLOATS = (LFS_RED_ADDTID ? 16 * RED_ROW : 64 * RED_STRIDE); // per wavefront
template <int ACC = 0>
LFS_DI void wave_sum16_atomic_lds(const v2f (&V)[8], float* __restrict__ dst, const uint32_t lane, float* __restrict__ scratch /* this wavefront's [RED_SCRATCH_FLOATS] */,
                                  unsigned long long* __restrict__ det64 = nullptr) {
    float c[16];
#if LFS_RED_ADDTID && !defined(LFS_EMULATE)
    {
        const uint32_t base = __builtin_amdgcn_readfirstlane(uint32_t(reinterpret_cast<uintptr_t>(scratch))); // LDS byte address (low half of the flat address)
        // M0 is a register the compiler manages itself (LLVM does not promise to honour an "m0" clobber): the block saves it, sets it, and puts it back - it leaves no
        // trace in M0, so nothing depends on how the compiler places its own M0 initialisations around the asm. (s_nop: one wait state between the SALU write
        // of M0 and an add-TID LDS instruction; the stores have read M0 when they issue, so the restore needs none.)
        uint32_t m0_saved;
        asm volatile("s_mov_b32 %[sv], m0\n\ts_mov_b32 m0, %[base]\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %[a0] offset:0\n\tds_write_addtid_b32 %[a1] offset:288\n\tds_write_addtid_b32 %[a2] offset:576\n\tds_write_addtid_b32 %[a3] offset:864\n\t"
                     "ds_write_addtid_b32 %[a4] offset:1152\n\tds_write_addtid_b32 %[a5] offset:1440\n\tds_write_addtid_b32 %[a6] offset:1728\n\tds_write_addtid_b32 %[a7] offset:2016\n\t"
                     "ds_write_addtid_b32 %[a8] offset:2304\n\tds_write_addtid_b32 %[a9] offset:2592\n\tds_write_addtid_b32 %[a10] offset:2880\n\tds_write_addtid_b32 %[a11] offset:3168\n\t"
                     "ds_write_addtid_b32 %[a12] offset:3456\n\tds_write_addtid_b32 %[a13] offset:3744\n\tds_write_addtid_b32 %[a14] offset:4032\n\tds_write_addtid_b32 %[a15] offset:4320\n\t"
                     "s_mov_b32 m0, %[sv]"
                     : [sv] "=&s"(m0_saved)
                     : [a0] "v"(V[0].x), [a1] "v"(V[0].y), [a2] "v"(V[1].x), [a3] "v"(V[1].y), [a4] "v"(V[2].x), [a5] "v"(V[2].y), [a6] "v"(V[3].x), [a7] "v"(V[3].y),
                       [a8] "v"(V[4].x), [a9] "v"(V[4].y), [a10] "v"(V[5].x), [a11] "v"(V[5].y), [a12] "v"(V[6].x), [a13] "v"(V[6].y), [a14] "v"(V[7].x), [a15] "v"(V[7].y),
                       [base] "s"(base) : "memory");
        __builtin_amdgcn_wave_barrier();
        const float4* rd = reinterpret_cast<const float4*>(scratch + (lane & 15) * RED_ROW + 4 * (lane >> 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 r = rd[4 * j]; c[4 * j] = r.x; c[4 * j + 1] = r.y; c[4 * j + 2] = r.z; c[4 * j + 3] = r.w; }
        __builtin_amdgcn_wave_barrier();
    }
#else
    float* wr = scratch + lane * RED_STRIDE;
#pragma unroll
    for (int j = 0; j < 8; ++j) { wr[2 * j] = V[j].x; wr[2 * j + 1] = V[j].y; }
    LFS_WAVE_LOCKSTEP();
#ifndef LFS_EMULATE
    __builtin_amdgcn_wave_barrier(); // (no instruction: keeps the compiler from moving the reads across the writes of OTHER lanes it cannot see)
#endif
    const float* rd = scratch + (lane >> 4) * (16 * RED_STRIDE) + (lane & 15);
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = rd[i * RED_STRIDE];
    LFS_WAVE_LOCKSTEP();
#ifndef LFS_EMULATE
    __builtin_amdgcn_wave_barrier();
#endif
#endif
    v2f p0 = v2f{c[0], c[1]} + v2f{c[2], c[3]}, p1 = v2f{c[4], c[5]} + v2f{c[6], c[7]}, p2 = v2f{c[8], c[9]} + v2f{c[10], c[11]}, p3 = v2f{c[12], c[13]} + v2f{c[14], c[15]};
    p0 += p1; p2 += p3; p0 += p2;
    float t = p0.x + p0.y;
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    if (lane < 16) { // lane L holds the total of v[L]
        if (ACC == 0) unsafeAtomicAdd(dst + lane, t);
#ifndef LFS_EMULATE
        else if (ACC == 1) atomicMax(reinterpret_cast<uint32_t*>(dst) + lane, __float_as_uint(t) & 0x7fffffffu);
        else {
            const uint32_t mbits = reinterpret_cast<const uint32_t*>(dst)[lane];
            if (mbits != 0u && t != 0.f) {
                const int e = max(int((mbits >> 23) & 0xffu), 1) - 127;
                atomicAdd(det64 + lane, (unsigned long long)__float2ll_rn(ldexpf(t, 40 - e)));
            }
        }
#endif
    }
}

// LFS_RED_QUAD (round 6): the same value-major block, read so that the four partial sums of a slot land in ONE QUAD: lane L (k = L >> 2, q = L & 3) reads the pieces
// {16 j + 4 q .. + 3}, j = 0..3, of row k, and the cross-lane part of the reduction is two quad-permute DPP adds instead of mov + v_permlane16_swap + add + mov +
// v_permlane32_swap + add (6 VALU and their wait states). Row stride 80 floats: the 8 lanes of a ds_read_b128 service group (two rows x four quarters) hit 8 disjoint 4-bank
// ranges ((16 k + 4 q) mod 64). The atomic goes out with the accumulator row's address in an SGPR pair (the Gaussian is wave-uniform) and a constant per-lane offset: no
// 64-bit VALU add per evaluation. lds_base = the block's LDS byte address, read once per kernel (the compiler re-issued v_readfirstlane per evaluation).
#ifndef LFS_RED_QUAD
#define LFS_RED_QUAD 1
#endif
constexpr int RED_QROW = 80;
constexpr int RED_QUAD_SCRATCH_FLOATS = 16 * RED_QROW;
#if LFS_RED_QUAD && LFS_RED_ADDTID && !defined(LFS_EMULATE)   // (-DLFS_RED_ADDTID=0, the test suite's second build: compiler-generated stores in the round-3 layout)
#define LFS_RED_QUAD_ASM 1
#else
#define LFS_RED_QUAD_ASM 0
#endif
#if LFS_RED_QUAD_ASM
// SKIP_9_11 (LFS_ACC_SYM rows: slots 9 .. 11 carry nothing): their three stores are left out, their quads read whatever the block held and must not reach the atomic -
// the caller's `atomic_lane` ((lane & 3) == 0 and slot not in 9 .. 11) says which lanes do.
// LFS_RED_M0_ONCE: M0 (the base of the add-TID stores) is written ONCE by the kernel (raster_bwd_kernel's prologue) instead of saved / set / restored around every block of
// stores (3 SALU + a wait state per evaluation). Nothing else in that kernel touches M0 on gfx950 (DS instructions do not need it since GFX9) - the compiler does not know
// about the asm's use of it, so tests/test_kernel_resources.py holds that statement against the disassembly of the shipped kernels (exactly one write of m0).
#ifndef LFS_RED_M0_ONCE
#define LFS_RED_M0_ONCE 1
#endif
#if LFS_RED_M0_ONCE
#define LFS_RED_M0_PROLOGUE "; m0 = %[base] (set in the kernel prologue), scratch %[sv]\n\t"
#define LFS_RED_M0_EPILOGUE ""
#else
#define LFS_RED_M0_PROLOGUE "s_mov_b32 %[sv], m0\n\ts_mov_b32 m0, %[base]\n\ts_nop 0\n\t"
#define LFS_RED_M0_EPILOGUE "s_mov_b32 m0, %[sv]"
#endif
// LFS_RED_BUF_ATOMIC (ACC == 0): the 13 / 16 totals leave through ONE buffer atomic with no EXEC round trip and no 64-bit address arithmetic. The accumulator is
// addressed as a raw buffer (descriptor built once per kernel: base = acc, num_records = its size in bytes); the row of the Gaussian is the instruction's SGPR offset
// (the SAME e.x << 6 the record load uses: rows and records are both 64 B), the lane's slot its VGPR offset - and a lane that carries no total (three of every quad,
// slots 9 .. 11 of an LFS_ACC_SYM row) holds RED_BUF_DEAD there, which the hardware's range check drops: 0x80000000 is beyond every accumulator this library accepts
// (C * N < 2^25 rows on this path, raster_check) whether the check adds the SGPR offset or not, and the sum does not wrap. Against `if (atomic_lane) global_atomic`:
// s_and_saveexec + s_cbranch_execz + s_lshl_b64 + s_add_u32 + s_addc_u32 + s_or exec -> nothing (tools/valu_rate.hip: an EXEC save / restore pair costs the SIMD as
// much as four v_fma_f32, and scalar instructions share one issue port per CU). The deterministic passes (ACC 1 / 2) keep the branch.
#ifndef LFS_RED_BUF_ATOMIC
#define LFS_RED_BUF_ATOMIC 1
#endif
constexpr uint32_t RED_BUF_DEAD = 0x80000000u;
static_assert(ACC_STRIDE * sizeof(float) == 64, "the buffer atomic's row offset is the record walker's e.x << 6");
#if LFS_RED_BUF_ATOMIC
struct RedBuf { __amdgpu_buffer_rsrc_t rsrc; uint32_t voff; };   // voff: 4 x slot on the lanes that add, RED_BUF_DEAD on the others
LFS_DI RedBuf red_buf_make(float* acc, const uint64_t rows, const uint32_t lane, const bool atomic_lane) {
    RedBuf b;
    b.rsrc = __builtin_amdgcn_make_buffer_rsrc(acc, 0, uint32_t(rows * uint64_t(ACC_STRIDE * sizeof(float))), 0x00020000);   // raw buffer, 32-bit data format (the gfx9 word 3 of every untyped buffer)
    b.voff = atomic_lane ? lane : RED_BUF_DEAD;   // lane = 4 x slot on the first lane of a quad
    return b;
}
#endif
template <int ACC = 0, bool SKIP_9_11 = false>
LFS_DI void wave_sum16_atomic_quad(const v2f (&V)[8], float* __restrict__ dst /* wave-uniform */, const uint32_t lane, const uint32_t lds_base, const float4* __restrict__ rd /* this lane's read pointer */,
                                   const bool atomic_lane, unsigned long long* __restrict__ det64 = nullptr
#if LFS_RED_BUF_ATOMIC
                                   , const RedBuf* __restrict__ rb = nullptr, const uint32_t row_bytes = 0u /* wave-uniform: 64 x the Gaussian's row */
#endif
                                   ) {
    float c[16];
    {
        uint32_t m0_saved; // (M0 saved and put back inside the block: see wave_sum16_atomic_lds)
        if (SKIP_9_11)
        asm volatile(LFS_RED_M0_PROLOGUE
                     "ds_write_addtid_b32 %[a0] offset:0\n\tds_write_addtid_b32 %[a1] offset:320\n\tds_write_addtid_b32 %[a2] offset:640\n\tds_write_addtid_b32 %[a3] offset:960\n\t"
                     "ds_write_addtid_b32 %[a4] offset:1280\n\tds_write_addtid_b32 %[a5] offset:1600\n\tds_write_addtid_b32 %[a6] offset:1920\n\tds_write_addtid_b32 %[a7] offset:2240\n\t"
                     "ds_write_addtid_b32 %[a8] offset:2560\n\t"
                     "ds_write_addtid_b32 %[a12] offset:3840\n\tds_write_addtid_b32 %[a13] offset:4160\n\tds_write_addtid_b32 %[a14] offset:4480\n\tds_write_addtid_b32 %[a15] offset:4800\n\t"
                     LFS_RED_M0_EPILOGUE
                     : [sv] "=&s"(m0_saved)
                     : [a0] "v"(V[0].x), [a1] "v"(V[0].y), [a2] "v"(V[1].x), [a3] "v"(V[1].y), [a4] "v"(V[2].x), [a5] "v"(V[2].y), [a6] "v"(V[3].x), [a7] "v"(V[3].y),
                       [a8] "v"(V[4].x), [a12] "v"(V[6].x), [a13] "v"(V[6].y), [a14] "v"(V[7].x), [a15] "v"(V[7].y),
                       [base] "s"(lds_base) : "memory");
        else
        asm volatile(LFS_RED_M0_PROLOGUE
                     "ds_write_addtid_b32 %[a0] offset:0\n\tds_write_addtid_b32 %[a1] offset:320\n\tds_write_addtid_b32 %[a2] offset:640\n\tds_write_addtid_b32 %[a3] offset:960\n\t"
                     "ds_write_addtid_b32 %[a4] offset:1280\n\tds_write_addtid_b32 %[a5] offset:1600\n\tds_write_addtid_b32 %[a6] offset:1920\n\tds_write_addtid_b32 %[a7] offset:2240\n\t"
                     "ds_write_addtid_b32 %[a8] offset:2560\n\tds_write_addtid_b32 %[a9] offset:2880\n\tds_write_addtid_b32 %[a10] offset:3200\n\tds_write_addtid_b32 %[a11] offset:3520\n\t"
                     "ds_write_addtid_b32 %[a12] offset:3840\n\tds_write_addtid_b32 %[a13] offset:4160\n\tds_write_addtid_b32 %[a14] offset:4480\n\tds_write_addtid_b32 %[a15] offset:4800\n\t"
                     LFS_RED_M0_EPILOGUE
                     : [sv] "=&s"(m0_saved)
                     : [a0] "v"(V[0].x), [a1] "v"(V[0].y), [a2] "v"(V[1].x), [a3] "v"(V[1].y), [a4] "v"(V[2].x), [a5] "v"(V[2].y), [a6] "v"(V[3].x), [a7] "v"(V[3].y),
                       [a8] "v"(V[4].x), [a9] "v"(V[4].y), [a10] "v"(V[5].x), [a11] "v"(V[5].y), [a12] "v"(V[6].x), [a13] "v"(V[6].y), [a14] "v"(V[7].x), [a15] "v"(V[7].y),
                       [base] "s"(lds_base) : "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 r = rd[4 * j]; c[4 * j] = r.x; c[4 * j + 1] = r.y; c[4 * j + 2] = r.z; c[4 * j + 3] = r.w; }
        __builtin_amdgcn_wave_barrier();
    }
    v2f p0 = v2f{c[0], c[1]} + v2f{c[2], c[3]}, p1 = v2f{c[4], c[5]} + v2f{c[6], c[7]}, p2 = v2f{c[8], c[9]} + v2f{c[10], c[11]}, p3 = v2f{c[12], c[13]} + v2f{c[14], c[15]};
    p0 += p1; p2 += p3; p0 += p2;
    float t = p0.x + p0.y;
    t += dpp_mov<0xB1>(t);   // lane ^ 1
    t += dpp_mov<0x4E>(t);   // lane ^ 2: every lane of quad k holds the total of slot k
#if LFS_RED_BUF_ATOMIC
    if (ACC == 0) {
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(t, rb->rsrc, int(rb->voff), int(row_bytes), 0);   // buffer_atomic_add_f32 v, v, s[4], s offen - every lane issues, the dead ones are out of range
        return;
    }
#endif
    asm volatile("" : "+v"(t)); // (no instruction: keeps the second add in front of the one-lane-in-four branch, where it folds into a v_add_f32_dpp; sunk into the branch it is mov + mov_dpp + add)
    if (atomic_lane) {
        const uint32_t slot = lane >> 2;
        if (ACC == 0) {
            const uint32_t voff = lane;   // = 4 bytes x slot
            asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(voff), "v"(t), "s"(dst) : "memory");
        } else if (ACC == 1) atomicMax(reinterpret_cast<uint32_t*>(dst) + slot, __float_as_uint(t) & 0x7fffffffu);
        else {
            const uint32_t mbits = reinterpret_cast<const uint32_t*>(dst)[slot];
            if (mbits != 0u && t != 0.f) {
                const int e = max(int((mbits >> 23) & 0xffu), 1) - 127;
                atomicAdd(det64 + slot, (unsigned long long)__float2ll_rn(ldexpf(t, 40 - e)));
            }
        }
    }
}
#endif

// The EWA blend backward's NINE sums the same way (a [64][9] block: the odd stride is conflict-free for the row writes and for the column reads alike);
// lanes 9..15 of every quarter read a duplicate column and are dropped at the atomic. Replaces wave_sum8_atomic + wave_sum1 + two atomics.
constexpr int RED9_STRIDE = 9;
constexpr int RED9_SCRATCH_FLOATS = (LFS_RED_ADDTID ? 16 * RED_ROW : 64 * RED9_STRIDE); // per wavefront (value-major form: rows 9..15 are read by the dropped lanes, never written)
LFS_DI void wave_sum9_atomic_lds(const float (&v)[9], float* __restrict__ dst, const uint32_t lane, float* __restrict__ scratch /* this wavefront's [RED9_SCRATCH_FLOATS] */) {
    float c[16];
#if LFS_RED_ADDTID && !defined(LFS_EMULATE)
    {   // value-major block through ds_write_addtid_b32 / ds_read_b128, as wave_sum16_atomic_lds
        const uint32_t base = __builtin_amdgcn_readfirstlane(uint32_t(reinterpret_cast<uintptr_t>(scratch)));
        uint32_t m0_saved; // (M0 saved and put back inside the block: see wave_sum16_atomic_lds)
        asm volatile("s_mov_b32 %[sv], m0\n\ts_mov_b32 m0, %[base]\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %[a0] offset:0\n\tds_write_addtid_b32 %[a1] offset:288\n\tds_write_addtid_b32 %[a2] offset:576\n\tds_write_addtid_b32 %[a3] offset:864\n\t"
                     "ds_write_addtid_b32 %[a4] offset:1152\n\tds_write_addtid_b32 %[a5] offset:1440\n\tds_write_addtid_b32 %[a6] offset:1728\n\tds_write_addtid_b32 %[a7] offset:2016\n\t"
                     "ds_write_addtid_b32 %[a8] offset:2304\n\t"
                     "s_mov_b32 m0, %[sv]"
                     : [sv] "=&s"(m0_saved)
                     : [a0] "v"(v[0]), [a1] "v"(v[1]), [a2] "v"(v[2]), [a3] "v"(v[3]), [a4] "v"(v[4]), [a5] "v"(v[5]), [a6] "v"(v[6]), [a7] "v"(v[7]), [a8] "v"(v[8]),
                       [base] "s"(base) : "memory");
        __builtin_amdgcn_wave_barrier();
        const float4* rd = reinterpret_cast<const float4*>(scratch + (lane & 15) * RED_ROW + 4 * (lane >> 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 r = rd[4 * j]; c[4 * j] = r.x; c[4 * j + 1] = r.y; c[4 * j + 2] = r.z; c[4 * j + 3] = r.w; }
        __builtin_amdgcn_wave_barrier();
    }
#else
    float* wr = scratch + lane * RED9_STRIDE;
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = v[k];
    LFS_WAVE_LOCKSTEP();
#ifndef LFS_EMULATE
    __builtin_amdgcn_wave_barrier();
#endif
    const uint32_t r = min(lane & 15u, 8u);
    const float* rd = scratch + (lane >> 4) * (16 * RED9_STRIDE) + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = rd[i * RED9_STRIDE];
    LFS_WAVE_LOCKSTEP();
#ifndef LFS_EMULATE
    __builtin_amdgcn_wave_barrier();
#endif
#endif
    v2f p0 = v2f{c[0], c[1]} + v2f{c[2], c[3]}, p1 = v2f{c[4], c[5]} + v2f{c[6], c[7]}, p2 = v2f{c[8], c[9]} + v2f{c[10], c[11]}, p3 = v2f{c[12], c[13]} + v2f{c[14], c[15]};
    p0 += p1; p2 += p3; p0 += p2;
    float t = p0.x + p0.y;
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    if (lane < 9) unsafeAtomicAdd(dst + lane, t);
}

// 8 per-lane values -> 8 totals with one 8-lane atomic instruction (same halving scheme as wave_sum16_atomic: 18 VALU), and a
// single value -> its total by a row butterfly + two cross-row folds (7 VALU). Used by the EWA blend backward (9 sums).
LFS_DI void wave_sum8_atomic(const float (&v)[8], float* __restrict__ dst, const uint32_t lane) {
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 4]), false, false);
        w[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    float u[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[j]), __float_as_uint(w[j + 2]), false, false);
        u[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // row r (16 lanes) holds partial sums of v[j + 2r] in u[j]
    const bool b0 = lane & 1;
    float t = (b0 ? u[1] : u[0]) + dpp_mov<0xB1>(b0 ? u[0] : u[1]);
    t += dpp_mov<0x4E>(t);
    t += dpp_mov<0x124>(t);
    t += dpp_mov<0x128>(t);
    if ((lane & 14) == 0) unsafeAtomicAdd(dst + 2 * (lane >> 4) + (lane & 1), t);
}
LFS_DI float wave_sum1(float v) { // every lane gets the total
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

} // namespace lfs

// Microbenchmark (round 6): VALU / SALU issue cost per instruction FORM on gfx950, as cycles of one SIMD per wave64 instruction, at 1 .. 8 wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > profiles/r06/valu_rate.json
// Why: tools/pk_rate.hip (round 1) measured 4.4 cycles for v_fma_f32 and the guide (MI355X_MICROARCH.md, "Per-instruction cycle constants") says 2; the rasterizer's
// "VALU-issue-bound at 90 %" rests on the 4. Every body is 32 volatile inline-asm instructions on 16 independent accumulators (no dependent pair closer than 16
// instructions), timed with s_memtime inside the wavefront; cycles per instruction of ONE SIMD = ticks x (s_memtime tick / shader cycle) / (instructions x wavefronts on the
// SIMD). s_memtime counts at a fixed 100 MHz-derived rate on some parts and at the shader clock on others: the kernel time from HIP events is printed beside it, so the two
// can be reconciled (ns per instruction per SIMD is clock-free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef float v2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Form { FMAC_VOP2_SGPR, FMA_VOP3_SGPR, FMA_VOP3_VGPR, MUL_VOP2_SGPR, ADD_VOP2, PK_FMA_VGPR, PK_FMA_SGPR, PK_MUL_SGPR, EXP, RCP, MIN_VOP2_LIT, SALU_MOV, VALU_SALU_1_1, VALU_SALU_2_1, CNDMASK, CNDMASK_E64_SGPR, CMP_VCC, CMP_E64_SGPR, CMP_THEN_CNDMASK, AND_B32, MOV_B32, READFIRSTLANE, SUB_VOP2, MAX_VOP2, FMA_NEG_VOP3, PERMLANE32_SWAP, PERMLANE16_SWAP, ADD_DPP_ROR, FMAC_HALF_EXEC, SAVEEXEC, BALLOT_BRANCH, N_FORMS };
static const char* form_name[N_FORMS] = {"v_fmac_f32_e32 v,s,v", "v_fma_f32 v,v,s,v", "v_fma_f32 v,v,v,v", "v_mul_f32_e32 v,s,v", "v_add_f32_e32 v,v,v", "v_pk_fma_f32 v,v,v,v", "v_pk_fma_f32 v,s,v,v",
                                         "v_pk_mul_f32 v,s,v", "v_exp_f32", "v_rcp_f32", "v_min_f32_e32 v,lit,v", "s_mov_b32 (32 per body)", "v_fmac + s_mov alternating (32 + 32)", "v_fmac x2 + s_mov (32 + 16)", "v_cndmask_b32_e32 v,v,v,vcc (vcc never written)", "v_cndmask_b32_e64 v,0,v,s[a:b]", "v_cmp_lt_f32_e32 vcc,v,v", "v_cmp_lt_f32_e64 s[a:b],v,v", "v_cmp_e32 + v_cndmask_e32 pairs (16 + 16, counted 32)", "v_and_b32_e32", "v_mov_b32_e32", "v_readfirstlane_b32", "v_sub_f32_e32 v,1.0,v", "v_max_f32_e32", "v_fma_f32 v,-v,v,s", "v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_add_f32_dpp row_ror:4", "v_fmac_f32_e32 with exec = low 32 lanes", "s_and_saveexec_b64 + s_or_b64 exec (16 + 16, counted 32 SALU)", "v_cmp_e32 + s_cbranch_vccz not taken (16 + 16, counted 16 VALU)"};

template <int FORM>
__global__ void __launch_bounds__(256) k(unsigned long long* __restrict__ ticks, float* __restrict__ out, int iters, float sa, float sb) {
    float a[16]; v2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = float(threadIdx.x + i) * 1e-3f; p[i] = v2{a[i], a[i] + 1.f}; }
    const float b = float(threadIdx.x) * 1e-6f + 0.5f;
    const v2 pb{b, b};
    const v2 ps{sa, sb};
    unsigned int sdummy = 0;
    unsigned long long smask = 0x5555aaaa3333ccccull | (unsigned long long)(iters), smask2 = 0;
    if (FORM == FMAC_HALF_EXEC) asm volatile("s_mov_b64 exec, 0xffffffff" :::);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (FORM == FMAC_VOP2_SGPR) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(sa), "v"(b));
                REP16(X)
#undef X
            } else if (FORM == FMA_VOP3_SGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sa), "v"(b));
                REP16(X)
#undef X
            } else if (FORM == FMA_VOP3_VGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(b));
                REP16(X)
#undef X
            } else if (FORM == MUL_VOP2_SGPR) {
#define X(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(sa));
                REP16(X)
#undef X
            } else if (FORM == ADD_VOP2) {
#define X(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP16(X)
#undef X
            } else if (FORM == PK_FMA_VGPR) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
                REP16(X)
#undef X
            } else if (FORM == PK_FMA_SGPR) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "s"(ps), "v"(pb));
                REP16(X)
#undef X
            } else if (FORM == PK_MUL_SGPR) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "s"(ps));
                REP16(X)
#undef X
            } else if (FORM == EXP) {
#define X(i) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (FORM == RCP) {
#define X(i) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (FORM == MIN_VOP2_LIT) {
#define X(i) asm volatile("v_min_f32_e32 %0, 0x3f7fbe77, %0" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (FORM == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                REP16(X)
#undef X
            } else if (FORM == SALU_MOV) {
#define X(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy) : : "scc");
                REP16(X)
#undef X
            } else if (FORM == VALU_SALU_1_1) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %2, %3\n\ts_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(sdummy) : "s"(sa), "v"(b) : "scc");
                REP16(X)
#undef X
            } else if (FORM == VALU_SALU_2_1) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %2, %3" : "+v"(a[i]), "+s"(sdummy) : "s"(sa), "v"(b)); if ((i) & 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy) : : "scc");
                REP16(X)
#undef X

            } else if (FORM == CNDMASK_E64_SGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[i]) : "s"(smask));
                REP16(X)
#undef X
            } else if (FORM == CMP_VCC) {
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                REP16(X)
#undef X
            } else if (FORM == CMP_E64_SGPR) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(smask2) : "v"(a[i]), "v"(b));
                REP16(X)
#undef X
            } else if (FORM == CMP_THEN_CNDMASK) {
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                REP16(X)
#undef X
            } else if (FORM == AND_B32) {
#define X(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP16(X)
#undef X
            } else if (FORM == MOV_B32) {
#define X(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(b));
                REP16(X)
#undef X
            } else if (FORM == READFIRSTLANE) {
#define X(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sdummy) : "v"(a[i]));
                REP16(X)
#undef X
            } else if (FORM == SUB_VOP2) {
#define X(i) asm volatile("v_sub_f32_e32 %0, 1.0, %0" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (FORM == MAX_VOP2) {
#define X(i) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP16(X)
#undef X
            } else if (FORM == FMA_NEG_VOP3) {
#define X(i) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(sa));
                REP16(X)
#undef X
            } else if (FORM == PERMLANE32_SWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            } else if (FORM == PERMLANE16_SWAP) {
#define X(i) asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            } else if (FORM == ADD_DPP_ROR) {
#define X(i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (FORM == FMAC_HALF_EXEC) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(sa), "v"(b));
                REP16(X)
#undef X
            } else if (FORM == SAVEEXEC) {
#define X(i) asm volatile("s_and_saveexec_b64 %0, %1\n\ts_or_b64 exec, exec, %0" : "=&s"(smask2) : "s"(smask) : "scc");
                REP16(X)
#undef X
            } else if (FORM == BALLOT_BRANCH) {
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\ts_cbranch_vccz 0" : : "v"(a[i]), "v"(b) : "vcc");
                REP16(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (FORM == FMAC_HALF_EXEC) asm volatile("s_mov_b64 exec, -1" :::);
    float s = float(sdummy) + float(smask2 & 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int FORM>
static void run_form(int cus, unsigned long long* d_ticks, float* d_out, bool& first) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 7, 8}) {           // wavefronts per SIMD = workgroups of 256 threads per CU
        const int grid = cus * wps;
        hipLaunchKernelGGL((k<FORM>), dim3(grid), dim3(256), 0, 0, d_ticks, d_out, 10, 0.999f, 1.001f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<FORM>), dim3(grid), dim3(256), 0, 0, d_ticks, d_out, iters, 0.999f, 1.001f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(size_t(grid) * 4);
        (void)hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost);
        double tsum = 0; for (auto t : h) tsum += double(t);
        const double ticks = tsum / double(h.size());
        const double valu = (FORM == SALU_MOV || FORM == SAVEEXEC) ? 0.0 : (FORM == BALLOT_BRANCH ? 16.0 : 32.0), salu = (FORM == SALU_MOV || FORM == SAVEEXEC) ? 32.0 : (FORM == VALU_SALU_1_1 ? 32.0 : (FORM == VALU_SALU_2_1 ? 16.0 : 0.0));
        const double n = double(iters) * (valu > 0 ? valu : salu); // counted instructions per wavefront (VALU when there are any)
        printf("%s  {\"form\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"ns_per_instr_per_simd\": %.3f, \"memtime_ticks_per_instr_per_simd\": %.3f}", first ? "" : ",\n",
               form_name[FORM], wps, ms, ms * 1e6 / (n * wps), ticks / (n * wps));
        first = false;
    }
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_ticks; float* d_out;
    (void)hipMalloc(&d_ticks, 8 * size_t(cus) * 8 * 4);
    (void)hipMalloc(&d_out, 4 * size_t(cus) * 8 * 256);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"note\": \"ns_per_instr_per_simd x clock = cycles; at 2.4 GHz 1 ns = 2.4 cycles\",\n \"rows\": [\n", prop.name, cus, prop.clockRate / 1000);
    bool first = true;
    run_form<FMAC_VOP2_SGPR>(cus, d_ticks, d_out, first);
    run_form<FMA_VOP3_SGPR>(cus, d_ticks, d_out, first);
    run_form<FMA_VOP3_VGPR>(cus, d_ticks, d_out, first);
    run_form<MUL_VOP2_SGPR>(cus, d_ticks, d_out, first);
    run_form<ADD_VOP2>(cus, d_ticks, d_out, first);
    run_form<PK_FMA_VGPR>(cus, d_ticks, d_out, first);
    run_form<PK_FMA_SGPR>(cus, d_ticks, d_out, first);
    run_form<PK_MUL_SGPR>(cus, d_ticks, d_out, first);
    run_form<EXP>(cus, d_ticks, d_out, first);
    run_form<RCP>(cus, d_ticks, d_out, first);
    run_form<MIN_VOP2_LIT>(cus, d_ticks, d_out, first);
    run_form<CNDMASK>(cus, d_ticks, d_out, first);
    run_form<SALU_MOV>(cus, d_ticks, d_out, first);
    run_form<VALU_SALU_1_1>(cus, d_ticks, d_out, first);
    run_form<VALU_SALU_2_1>(cus, d_ticks, d_out, first);
    run_form<CNDMASK_E64_SGPR>(cus, d_ticks, d_out, first);
    run_form<CMP_VCC>(cus, d_ticks, d_out, first);
    run_form<CMP_E64_SGPR>(cus, d_ticks, d_out, first);
    run_form<CMP_THEN_CNDMASK>(cus, d_ticks, d_out, first);
    run_form<AND_B32>(cus, d_ticks, d_out, first);
    run_form<MOV_B32>(cus, d_ticks, d_out, first);
    run_form<READFIRSTLANE>(cus, d_ticks, d_out, first);
    run_form<SUB_VOP2>(cus, d_ticks, d_out, first);
    run_form<MAX_VOP2>(cus, d_ticks, d_out, first);
    run_form<FMA_NEG_VOP3>(cus, d_ticks, d_out, first);
    run_form<PERMLANE32_SWAP>(cus, d_ticks, d_out, first);
    run_form<PERMLANE16_SWAP>(cus, d_ticks, d_out, first);
    run_form<ADD_DPP_ROR>(cus, d_ticks, d_out, first);
    run_form<FMAC_HALF_EXEC>(cus, d_ticks, d_out, first);
    run_form<SAVEEXEC>(cus, d_ticks, d_out, first);
    run_form<BALLOT_BRANCH>(cus, d_ticks, d_out, first);
    printf("\n ]}\n");
    return 0;
}

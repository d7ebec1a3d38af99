// probe_valu.hip -- classify gfx950 VALU instruction forms by issue rate (wave64).
// Every kernel runs ITERS iterations of 8 independent copies of one instruction form at
// 8 waves/SIMD on all CUs; rate is reported as cycles per wave-instruction per SIMD at the
// clock measured in-kernel (s_memtime / s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_valu.hip -o tools/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)

#define DEFK(NAME, STMT)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, int iters, unsigned long long *clk) \
    {                                                                                                      \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        float va = a + threadIdx.x * 0.0f, vb = b + threadIdx.x * 0.0f;                                   \
        unsigned u = threadIdx.x * 2654435761u;                                                            \
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(u));                                                    \
        for (int i = 0; i < iters; ++i) { REP8(STMT) }                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                         \
            clk[0] = __builtin_amdgcn_s_memtime() - c0;                                                    \
            clk[1] = __builtin_amdgcn_s_memrealtime() - r0;                                                \
        }                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + __uint_as_float(u); \
    }

#define S_FMA_VVV(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(va), "v"(vb));
#define S_FMAC_VV(X) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(X) : "v"(va), "v"(vb));
#define S_FMAC_SV(X) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(X) : "s"(a), "v"(vb));
#define S_FMA_SVV(X) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(X) : "s"(a), "v"(vb));
#define S_FMA_VVV_ACC(X) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(X) : "v"(va), "v"(vb));
#define S_MUL_VV(X) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(X) : "v"(va));
#define S_MUL_SV(X) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(X) : "s"(a));
#define S_ADD_VV(X) asm volatile("v_add_f32 %0, %0, %1" : "+v"(X) : "v"(va));
#define S_MOV(X) asm volatile("v_mov_b32 %0, %1" : "=v"(X) : "v"(va));
#define S_AND(X) asm volatile("v_and_b32 %0, %0, %1" : "+v"(X) : "v"(u));
#define S_LSHL(X) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(X));
#define S_ADDU(X) asm volatile("v_add_u32 %0, %0, %1" : "+v"(X) : "v"(u));
#define S_CVT_U32(X) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(X) : "v"(u));
#define S_CVT_UB0(X) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(X) : "v"(u));
#define S_CVT_UB3(X) asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(X) : "v"(u));
#define S_MED3(X) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(X) : "v"(va), "v"(vb));
#define S_MAX(X) asm volatile("v_max_f32 %0, %0, %1" : "+v"(X) : "v"(va));
#define S_PERM(X) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(X) : "v"(va), "v"(u));
#define S_BFE(X) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(X) : "v"(u));
#define S_CVTPK(X) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(X) : "v"(va));
#define S_DPP(X) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(X) : "v"(va));
#define S_DPP_ROW(X) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(X) : "v"(va));
#define S_FMAC_DPP(X) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(X) : "v"(va), "v"(vb));
#define S_MIX_LO(X) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(X) : "v"(u), "v"(va));
#define S_CVT_F16(X) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(X) : "v"(u));
#define S_LERP(X) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(X) : "v"(u), "v"(va));
#define S_DOT2(X) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(X) : "v"(u), "v"(va));
#define S_DOT2C(X) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(X) : "v"(u), "v"(va));
#define S_SDWA_MUL(X) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(X) : "v"(va), "v"(vb));

DEFK(k_fma_vvv, S_FMA_VVV)
DEFK(k_fma_vvv_acc, S_FMA_VVV_ACC)
DEFK(k_fmac_vv, S_FMAC_VV)
DEFK(k_fmac_sv, S_FMAC_SV)
DEFK(k_fma_svv, S_FMA_SVV)
DEFK(k_mul_vv, S_MUL_VV)
DEFK(k_mul_sv, S_MUL_SV)
DEFK(k_add_vv, S_ADD_VV)
DEFK(k_mov, S_MOV)
DEFK(k_and, S_AND)
DEFK(k_lshl, S_LSHL)
DEFK(k_addu, S_ADDU)
DEFK(k_cvt_u32, S_CVT_U32)
DEFK(k_cvt_ub0, S_CVT_UB0)
DEFK(k_cvt_ub3, S_CVT_UB3)
DEFK(k_med3, S_MED3)
DEFK(k_max, S_MAX)
DEFK(k_perm, S_PERM)
DEFK(k_bfe, S_BFE)
DEFK(k_cvtpk, S_CVTPK)
DEFK(k_dpp, S_DPP)
DEFK(k_dpp_row, S_DPP_ROW)
DEFK(k_fmac_dpp, S_FMAC_DPP)
DEFK(k_mix_lo, S_MIX_LO)
DEFK(k_cvt_f16, S_CVT_F16)
DEFK(k_lerp, S_LERP)
DEFK(k_dot2, S_DOT2)
DEFK(k_dot2c, S_DOT2C)
DEFK(k_sdwa_mul, S_SDWA_MUL)

// packed-f32 forms: operands are VGPR pairs
typedef float f2 __attribute__((ext_vector_type(2)));
#define DEFK2(NAME, STMT)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, int iters, unsigned long long *clk) \
    {                                                                                                      \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        const float t = threadIdx.x;                                                                       \
        f2 x0 = {t, t + 1}, x1 = {t + 2, t + 3}, x2 = {t + 4, t + 5}, x3 = {t + 6, t + 7}, x4 = {t + 8, t + 9}, \
           x5 = {t + 10, t + 11}, x6 = {t + 12, t + 13}, x7 = {t + 14, t + 15};                            \
        f2 va = {a + t * 0.0f, a}, vb = {b + t * 0.0f, b};                                                 \
        asm volatile("" : "+v"(va), "+v"(vb));                                                             \
        for (int i = 0; i < iters; ++i) { REP8(STMT) }                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                         \
            clk[0] = __builtin_amdgcn_s_memtime() - c0;                                                    \
            clk[1] = __builtin_amdgcn_s_memrealtime() - r0;                                                \
        }                                                                                                  \
        const f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                            \
    }
#define S_PK_FMA(X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(X) : "v"(va), "v"(vb));
#define S_PK_FMA_BCAST(X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(X) : "v"(va), "v"(vb));
#define S_PK_MUL(X) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(X) : "v"(va));
#define S_PK_ADD(X) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(X) : "v"(va));
#define S_PK_MOV(X) asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(X) : "v"(va), "v"(vb));
DEFK2(k_pk_fma, S_PK_FMA)
DEFK2(k_pk_fma_bcast, S_PK_FMA_BCAST)
DEFK2(k_pk_mul, S_PK_MUL)
DEFK2(k_pk_add, S_PK_ADD)
DEFK2(k_pk_mov, S_PK_MOV)

// mixes: do the two rate classes overlap (time = max) or serialise (time = sum)?
#define REP4A(S) S(x0) S(x1) S(x2) S(x3)
#define REP4B(S) S(x4) S(x5) S(x6) S(x7)
#define DEFKMIX(NAME, SA, SB)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, int iters, unsigned long long *clk) \
    {                                                                                                      \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        float va = a + threadIdx.x * 0.0f, vb = b + threadIdx.x * 0.0f;                                   \
        unsigned u = threadIdx.x * 2654435761u;                                                            \
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(u));                                                    \
        for (int i = 0; i < iters; ++i) { SA(x0) SB(x4) SA(x1) SB(x5) SA(x2) SB(x6) SA(x3) SB(x7) }        \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                         \
            clk[0] = __builtin_amdgcn_s_memtime() - c0;                                                    \
            clk[1] = __builtin_amdgcn_s_memrealtime() - r0;                                                \
        }                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + __uint_as_float(u); \
    }
DEFKMIX(k_mix_fma_cvtpk, S_FMAC_VV, S_CVTPK)
DEFKMIX(k_mix_fma_dpp, S_FMAC_VV, S_DPP)
DEFKMIX(k_mix_fma_mulsv, S_FMAC_VV, S_MUL_SV)
DEFKMIX(k_mix_fma_fma, S_FMAC_VV, S_FMA_VVV_ACC)
DEFKMIX(k_mix_fma_dpprow, S_FMAC_VV, S_DPP_ROW)
DEFKMIX(k_mix_fma_cvtub, S_FMAC_VV, S_CVT_UB3)
DEFKMIX(k_mix_fma_perm, S_FMAC_VV, S_PERM)
DEFKMIX(k_mix_fma_lerp, S_FMAC_VV, S_LERP)
#define S_BPERM(X) asm volatile("ds_bpermute_b32 %0, %1, %2\n" : "=v"(X) : "v"(u), "v"(va));
DEFKMIX(k_mix_fma_bperm, S_FMAC_VV, S_BPERM)
#define S_FMAC_DPPW(X) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(X) : "v"(va), "v"(vb));
DEFKMIX(k_mix_fma_fmadpp, S_FMAC_VV, S_FMAC_DPPW)
DEFK(k_fmadppw, S_FMAC_DPPW)
#define S_FMAC2_DPPW(X) S_FMAC_VV(X) S_FMAC_VV(X) S_FMAC_DPPW(X) S_FMAC_VV(X)
DEFK(k_mix_3fma_1fmadpp, S_FMAC2_DPPW)
// dependent chains: 8 FMAs on ONE accumulator (ILP 1), on two (ILP 2), on four (ILP 4)
#define S_DEP1(X) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x0) : "v"(va), "v"(vb));
DEFK(k_dep1, S_DEP1)
#define REPDEP2 S_FMAC_VV(x0) S_FMAC_VV(x1) S_FMAC_VV(x0) S_FMAC_VV(x1) S_FMAC_VV(x0) S_FMAC_VV(x1) S_FMAC_VV(x0) S_FMAC_VV(x1)
#define S_DEP2(X) REPDEP2
#define REPDEP4 S_FMAC_VV(x0) S_FMAC_VV(x1) S_FMAC_VV(x2) S_FMAC_VV(x3) S_FMAC_VV(x0) S_FMAC_VV(x1) S_FMAC_VV(x2) S_FMAC_VV(x3)
// 3 fmac : 1 dpp
#define S_FMAC3_DPP(X) S_FMAC_VV(X) S_FMAC_VV(X) S_FMAC_VV(X) S_DPP(X)
DEFK(k_mix_3fma_1dpp, S_FMAC3_DPP)
__global__ __launch_bounds__(256) void k_dep2(float *out, float a, float b, int iters, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float va = a + threadIdx.x * 0.0f, vb = b + threadIdx.x * 0.0f;
    asm volatile("" : "+v"(va), "+v"(vb));
    for (int i = 0; i < iters; ++i) { REPDEP2 REPDEP2 REPDEP2 REPDEP2 REPDEP2 REPDEP2 REPDEP2 REPDEP2 }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ __launch_bounds__(256) void k_dep4(float *out, float a, float b, int iters, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float va = a + threadIdx.x * 0.0f, vb = b + threadIdx.x * 0.0f;
    asm volatile("" : "+v"(va), "+v"(vb));
    for (int i = 0; i < iters; ++i) { REPDEP4 REPDEP4 REPDEP4 REPDEP4 REPDEP4 REPDEP4 REPDEP4 REPDEP4 }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
#define S_FMAC3_CVTPK(X) S_FMAC_VV(X) S_FMAC_VV(X) S_FMAC_VV(X) S_CVTPK(X)
DEFK(k_mix_3fma_1cvtpk, S_FMAC3_CVTPK)

typedef void (*kern_t)(float *, float, float, int, unsigned long long *);

int main()
{
    float *o; CK(hipMalloc(&o, 2048 * 256 * 4));
    unsigned long long *clk; CK(hipMalloc(&clk, 16));
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32 x,x,v,v", k_fma_vvv}, {"v_fma_f32 x,v,v,x (acc)", k_fma_vvv_acc}, {"v_fmac_f32 x,v,v", k_fmac_vv},
        {"v_fmac_f32 x,s,v", k_fmac_sv}, {"v_fma_f32 x,s,v,x", k_fma_svv}, {"v_mul_f32 v,v", k_mul_vv}, {"v_mul_f32 s,v", k_mul_sv},
        {"v_add_f32 v,v", k_add_vv}, {"v_mov_b32", k_mov}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_add_u32", k_addu},
        {"v_cvt_f32_u32", k_cvt_u32}, {"v_cvt_f32_ubyte0", k_cvt_ub0}, {"v_cvt_f32_ubyte3", k_cvt_ub3}, {"v_med3_f32", k_med3},
        {"v_max_f32", k_max}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_cvt_pk_u8_f32", k_cvtpk},
        {"v_mov_b32_dpp wave_shr", k_dpp}, {"v_mov_b32_dpp row_shr", k_dpp_row}, {"v_fmac_f32_dpp row_shr", k_fmac_dpp},
        {"v_fma_mix_f32 f16lo", k_mix_lo}, {"v_cvt_f32_f16", k_cvt_f16}, {"v_lerp_u8", k_lerp}, {"v_dot2_f32_f16", k_dot2},
        {"v_dot2c_f32_f16", k_dot2c}, {"v_mul_f32_sdwa", k_sdwa_mul},
        {"v_pk_fma_f32 (2 fma)", k_pk_fma}, {"v_pk_fma_f32 bcast lo", k_pk_fma_bcast}, {"v_pk_mul_f32", k_pk_mul},
        {"v_pk_add_f32", k_pk_add}, {"v_pk_mov_b32", k_pk_mov},
        {"mix 4 fmac + 4 cvt_pk_u8", k_mix_fma_cvtpk}, {"mix 4 fmac + 4 mov_dpp", k_mix_fma_dpp},
        {"mix 4 fmac + 4 mul s,v", k_mix_fma_mulsv}, {"mix 4 fmac + 4 fma", k_mix_fma_fma},
        {"mix 4 fmac + 4 dpp row_shr", k_mix_fma_dpprow}, {"mix 4 fmac + 4 cvt_f32_ubyte3", k_mix_fma_cvtub},
        {"mix 4 fmac + 4 v_perm", k_mix_fma_perm}, {"mix 4 fmac + 4 v_lerp_u8", k_mix_fma_lerp},
        {"mix 4 fmac + 4 ds_bpermute", k_mix_fma_bperm}, {"mix 24 fmac + 8 dpp (x4 count)", k_mix_3fma_1dpp},
        {"mix 24 fmac + 8 cvt_pk (x4 count)", k_mix_3fma_1cvtpk},
        {"fmac dependent chain (ILP 1)", k_dep1}, {"fmac 2 chains (ILP 2)", k_dep2}, {"fmac 4 chains (ILP 4)", k_dep4},
        {"v_fmac_f32_dpp wave_shr", k_fmadppw}, {"mix 4 fmac + 4 fmac_dpp wave_shr", k_mix_fma_fmadpp},
        {"mix 24 fmac + 8 fmac_dpp (x4 count)", k_mix_3fma_1fmadpp}};
    const int iters = 20000;
    for (int occ : {2048, 768, 512, 256}) { // 8, 3, 2 and 1 waves per SIMD
        for (auto &e : ks) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(e.k, dim3(occ), dim3(256), 0, 0, o, 1.0001f, 0.5f, 1000, clk);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(occ), dim3(256), 0, 0, o, 1.0001f, 0.5f, iters, clk);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0;
            const double winstr_per_simd = 8.0 * iters * (occ / 256.0);
            const double cyc = ms * 1e-3 * ghz * 1e9 / winstr_per_simd;
            printf("[%d waves/SIMD] %-26s %7.3f ms  clock %.2f GHz  %.2f cycles per wave-instr per SIMD  (%.1f T lane-instr/s)\n",
                   occ / 256, e.name, ms, ghz, cyc, winstr_per_simd * 1024 * 64 / ms / 1e9);
        }
    }
    return 0;
}

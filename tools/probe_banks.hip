// probe_banks.hip -- do VGPR bank conflicts change the issue rate of v_fmac_f32 on gfx950?
// Registers are named explicitly; bank = register index mod 4 (assumption under test).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define BODY(D0,D1,D2,D3,D4,D5,D6,D7,A,B) \
    "v_fmac_f32 " D0 ", " A ", " B "\n v_fmac_f32 " D1 ", " A ", " B "\n v_fmac_f32 " D2 ", " A ", " B "\n v_fmac_f32 " D3 ", " A ", " B "\n" \
    "v_fmac_f32 " D4 ", " A ", " B "\n v_fmac_f32 " D5 ", " A ", " B "\n v_fmac_f32 " D6 ", " A ", " B "\n v_fmac_f32 " D7 ", " A ", " B "\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139"

#define DEFB(NAME, ASMBODY)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned long long *clk)          \
    {                                                                                                    \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        asm volatile("v_mov_b32 v132, 1.0\n v_mov_b32 v133, 0.5\n v_mov_b32 v134, 0.5\n v_mov_b32 v136, 0.5\n v_mov_b32 v137, 0.5\n v_mov_b32 v135, 0.5\n" ::: CLOB); \
        for (int i = 0; i < iters; ++i) asm volatile(ASMBODY ::: CLOB);                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                       \
            clk[0] = __builtin_amdgcn_s_memtime() - c0;                                                  \
            clk[1] = __builtin_amdgcn_s_memrealtime() - r0;                                              \
        }                                                                                                \
        float r;                                                                                         \
        asm volatile("v_mov_b32 %0, v100" : "=v"(r)::CLOB);                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                  \
    }
// dst banks 0 (v100,v104,...), src0 bank 1 (v133), src1 bank 2 (v134): no two operands share a bank
DEFB(k_nc, BODY("v100","v104","v108","v112","v116","v120","v124","v128","v133","v134"))
// src0 and src1 in the same bank (v133, v137: bank 1)
DEFB(k_c_src, BODY("v100","v104","v108","v112","v116","v120","v124","v128","v133","v137"))
// dst and src0 in the same bank (bank 0), src1 bank 2
DEFB(k_c_dst, BODY("v100","v104","v108","v112","v116","v120","v124","v128","v132","v134"))
// all three in bank 0
DEFB(k_c_all, BODY("v100","v104","v108","v112","v116","v120","v124","v128","v132","v136"))
// dst consecutive registers (banks 0,1,2,3,...), src banks 1 and 2: what a compiler typically produces
DEFB(k_seq, BODY("v100","v101","v102","v103","v104","v105","v106","v107","v133","v134"))
// same instruction twice on src0 == src1 register
DEFB(k_same, BODY("v100","v104","v108","v112","v116","v120","v124","v128","v133","v133"))

#define CLOBLO "v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23"
#define DEFR(NAME, ASMBODY, INIT, CL)                                                                    \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned long long *clk)          \
    {                                                                                                    \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        asm volatile(INIT ::: CL);                                                                       \
        for (int i = 0; i < iters; ++i) asm volatile(ASMBODY ::: CL);                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                       \
            clk[0] = __builtin_amdgcn_s_memtime() - c0;                                                  \
            clk[1] = __builtin_amdgcn_s_memrealtime() - r0;                                              \
        }                                                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)clk[0];                                      \
    }
DEFR(k_lo, BODY("v10","v11","v12","v13","v14","v15","v16","v17","v20","v21"), "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 0.5\n", CLOBLO)
DEFR(k_lo_b, BODY("v10","v11","v12","v13","v14","v15","v16","v17","v21","v22"), "v_mov_b32 v21, 1.0\n v_mov_b32 v22, 0.5\n", CLOBLO)
#define CLOBMID "v60","v61","v62","v63","v64","v65","v66","v67","v70","v71"
DEFR(k_mid, BODY("v60","v61","v62","v63","v64","v65","v66","v67","v70","v71"), "v_mov_b32 v70, 1.0\n v_mov_b32 v71, 0.5\n", CLOBMID)
DEFR(k_hi, BODY("v100","v101","v102","v103","v104","v105","v106","v107","v110","v111"), "v_mov_b32 v110, 1.0\n v_mov_b32 v111, 0.5\n", CLOB)
#define CLOBVH "v200","v201","v202","v203","v204","v205","v206","v207","v210","v211"
DEFR(k_vhi, BODY("v200","v201","v202","v203","v204","v205","v206","v207","v210","v211"), "v_mov_b32 v210, 1.0\n v_mov_b32 v211, 0.5\n", CLOBVH)
// dst and src1 in the same bank (dst v100.. bank 0..3 consecutive is k_hi; here dst bank 0, src1 bank 0, src0 bank 1)
DEFR(k_c_dst_src1, BODY("v100","v104","v108","v112","v116","v120","v124","v128","v133","v132"), "v_mov_b32 v133, 1.0\n v_mov_b32 v132, 0.5\n", CLOB)
// consecutive dsts, src0 bank 0 (conflicts with dst v100, v104), src1 bank 1
DEFR(k_seq_c0, BODY("v100","v101","v102","v103","v104","v105","v106","v107","v132","v133"), "v_mov_b32 v133, 1.0\n v_mov_b32 v132, 0.5\n", CLOB)
// v_fma_f32 (VOP3) with dst == src2, same conflict pattern as k_c_dst
#define BODY3(D0,D1,D2,D3,D4,D5,D6,D7,A,B) \
    "v_fma_f32 " D0 ", " A ", " B ", " D0 "\n v_fma_f32 " D1 ", " A ", " B ", " D1 "\n v_fma_f32 " D2 ", " A ", " B ", " D2 "\n v_fma_f32 " D3 ", " A ", " B ", " D3 "\n" \
    "v_fma_f32 " D4 ", " A ", " B ", " D4 "\n v_fma_f32 " D5 ", " A ", " B ", " D5 "\n v_fma_f32 " D6 ", " A ", " B ", " D6 "\n v_fma_f32 " D7 ", " A ", " B ", " D7 "\n"
DEFR(k_fma3_nc, BODY3("v100","v101","v102","v103","v104","v105","v106","v107","v110","v111"), "v_mov_b32 v110, 1.0\n v_mov_b32 v111, 0.5\n", CLOB)
// accumulate into the same 8 registers but sources far away
DEFR(k_lo_farsrc, BODY("v10","v11","v12","v13","v14","v15","v16","v17","v110","v111"), "v_mov_b32 v110, 1.0\n v_mov_b32 v111, 0.5\n", CLOB)

typedef void (*kern_t)(float *, int, unsigned long long *);
int main()
{
    float *o; CK(hipMalloc(&o, 2048 * 256 * 4));
    unsigned long long *clk; CK(hipMalloc(&clk, 16));
    struct { const char *name; kern_t k; } ks[] = {{"no bank shared (dst b0, src b1, b2)", k_nc}, {"src0, src1 same bank", k_c_src},
        {"dst, src0 same bank", k_c_dst}, {"all three same bank", k_c_all}, {"dst v100..v107 (banks 0-3), src b1, b2", k_seq},
        {"src0 == src1 register", k_same},
        {"dst v10-17, src v20,v21", k_lo}, {"dst v10-17, src v21,v22", k_lo_b}, {"dst v60-67, src v70,v71", k_mid},
        {"dst v100-107, src v110,v111", k_hi}, {"dst v200-207, src v210,v211", k_vhi}, {"dst v10-17, src v110,v111", k_lo_farsrc},
        {"dst b0, src0 b1, src1 b0 (dst/src1 same bank)", k_c_dst_src1}, {"dst v100-107, src0 b0, src1 b1", k_seq_c0},
        {"v_fma_f32 VOP3 dst v100-107 src v110,v111", k_fma3_nc}};
    const int iters = 20000;
    for (int occ : {768}) {
        for (auto &e : ks) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(e.k, dim3(occ), dim3(256), 0, 0, o, 1000, clk);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(occ), dim3(256), 0, 0, o, iters, clk);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0;
            const double winstr_per_simd = 8.0 * iters * (occ / 256.0);
            printf("[%d waves/SIMD] %-42s %7.3f ms  clock %.2f GHz  %.2f cycles per wave-instr per SIMD\n", occ / 256, e.name, ms, ghz,
                   ms * 1e-3 * ghz * 1e9 / winstr_per_simd);
        }
    }
    return 0;
}

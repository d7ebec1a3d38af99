// probe_swizzle.hip -- does gfx950 implement ds_swizzle_b32's rotate mode, and what does a lane exchange cost next
// to FMAs when it goes through the LDS crossbar (ds_swizzle, ds_bpermute) instead of the VALU (v_mov_b32_dpp)?
// Each timing kernel runs ITERS iterations of 12 independent v_fmac + 2 exchanges (the x2 Lanczos kernel's mix is
// about 6 : 1) at 3 waves per SIMD on all CUs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_swizzle.hip -o tools/probe_swizzle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_semantics(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_ds_swizzle(lane, 0xC000 | (0 << 10) | (1 << 5));       // rotate left by 1 within 32
    out[64 + lane] = __builtin_amdgcn_ds_swizzle(lane, 0xC000 | (1 << 10) | (1 << 5));  // rotate right by 1 within 32
}

#define FMA12 asm volatile("v_fmac_f32 %0, %12, %13\n v_fmac_f32 %1, %12, %13\n v_fmac_f32 %2, %12, %13\n v_fmac_f32 %3, %12, %13\n" \
                           "v_fmac_f32 %4, %12, %13\n v_fmac_f32 %5, %12, %13\n v_fmac_f32 %6, %12, %13\n v_fmac_f32 %7, %12, %13\n" \
                           "v_fmac_f32 %8, %12, %13\n v_fmac_f32 %9, %12, %13\n v_fmac_f32 %10, %12, %13\n v_fmac_f32 %11, %12, %13\n" \
                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), \
                             "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]) : "v"(va), "v"(vb));

template <int MODE> // 0: FMAs only, 1: + 2 dpp moves, 2: + 2 ds_swizzle rotate, 3: + 2 ds_bpermute
__global__ __launch_bounds__(256) void k_mix(float *out, float a, float b, int iters, unsigned long long *clk)
{
    __shared__ float pad[MODE == 99 ? 1 : 11 * 1024]; // 44 KB per block -> 3 blocks = 12 waves per CU
    pad[threadIdx.x] = a;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x + i;
    float va = a + threadIdx.x * 0.0f, vb = b + threadIdx.x * 0.0f;
    int e0 = threadIdx.x, e1 = threadIdx.x + 1;
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    asm volatile("" : "+v"(va), "+v"(vb), "+v"(e0), "+v"(e1));
    for (int i = 0; i < iters; ++i) {
        FMA12
        if (MODE == 1) {
            asm volatile("v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(e0), "=v"(e1) : "v"(x[0]), "v"(x[1]));
        } else if (MODE == 2) {
            e0 = __builtin_amdgcn_ds_swizzle(__float_as_int(x[0]), 0xC000 | (1 << 10) | (1 << 5));
            e1 = __builtin_amdgcn_ds_swizzle(__float_as_int(x[1]), 0xC000 | (0 << 10) | (1 << 5));
        } else if (MODE == 3) {
            e0 = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[0]));
            e1 = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[1]));
        }
        if (MODE != 0) { // consume the exchanged values a few FMAs later, as the kernel does
            x[6] += __int_as_float(e0) * 0.0f;
            x[7] += __int_as_float(e1) * 0.0f;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = pad[(threadIdx.x * 7) & 1023];
    for (int i = 0; i < 12; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, float *d_out, unsigned long long *d_clk)
{
    const int iters = 20000, blocks = 256 * 3;
    hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, 100, d_clk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, iters, d_clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CK(hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
    const double mhz = (double)clk[0] / ((double)clk[1] / 100.0); // memrealtime ticks at 100 MHz
    // 3 waves per SIMD share the SIMD: cycles per iteration per SIMD = time * clock / iters; per wave = that / 3
    const double cyc_iter = ms * 1e-3 * mhz * 1e6 / iters;
    printf("%-28s %8.3f ms  clock %6.0f MHz  %7.1f cycles per iteration per SIMD (3 waves)  = %6.1f per wave-iteration\n", name, ms, mhz, cyc_iter, cyc_iter / 3.0);
    return 0;
}

int main()
{
    int *d_sem;
    CK(hipMalloc(&d_sem, 128 * sizeof(int)));
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, d_sem);
    CK(hipDeviceSynchronize());
    std::vector<int> sem(128);
    CK(hipMemcpy(sem.data(), d_sem, 128 * sizeof(int), hipMemcpyDeviceToHost));
    printf("ds_swizzle 0xC020 (rotate, dir 0, by 1): lane 0 <- %d, lane 1 <- %d, lane 31 <- %d, lane 32 <- %d, lane 63 <- %d\n", sem[0], sem[1], sem[31], sem[32], sem[63]);
    printf("ds_swizzle 0xC420 (rotate, dir 1, by 1): lane 0 <- %d, lane 1 <- %d, lane 31 <- %d, lane 32 <- %d, lane 63 <- %d\n", sem[64], sem[65], sem[95], sem[96], sem[127]);
    float *d_out;
    unsigned long long *d_clk;
    CK(hipMalloc(&d_out, 256 * 3 * 256 * sizeof(float)));
    CK(hipMalloc(&d_clk, 2 * sizeof(unsigned long long)));
    if (run<0>("12 fmac", d_out, d_clk)) return 1;
    if (run<1>("12 fmac + 2 v_mov_dpp", d_out, d_clk)) return 1;
    if (run<2>("12 fmac + 2 ds_swizzle", d_out, d_clk)) return 1;
    if (run<3>("12 fmac + 2 ds_bpermute", d_out, d_clk)) return 1;
    return 0;
}

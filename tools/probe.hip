// probe.hip -- one-off hardware probes that decide kernel design choices (results are
// recorded in DESIGN.md).  Build: hipcc --offload-arch=gfx950 -O3 tools/probe.hip -o tools/probe
//  1. rounding of v_cvt_pk_u8_f32 (truncate? nearest-even? saturate?)
//  2. VALU issue rate: v_fma_f32 vs v_pk_fma_f32 vs separate mul+add, v_cvt_f32_ubyte
//  3. DPP wave_shr / wave_shl semantics across the 64 lanes
//  4. stream copy ceiling (16 B/lane read + write) for the roofline denominator
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_cvt(const float *in, unsigned *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}

__global__ void k_dpp(int *up, int *down)
{
    int l = threadIdx.x;
    up[l] = __builtin_amdgcn_update_dpp(-1, l, 0x138, 0xF, 0xF, false);
    down[l] = __builtin_amdgcn_update_dpp(-1, l, 0x130, 0xF, 0xF, false);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float *out, float a, float b, int iters, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 8 fma (asm: keeps the SLP vectoriser from packing them)
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) { // 4 pk_fma = 8 fma
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(av), "v"(bv));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(av), "v"(bv));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(av), "v"(bv));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(av), "v"(bv));
        } else if (MODE == 2) { // 8 mul + 8 add, unfused
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x0) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x1) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x2) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x3) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x4) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x5) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x6) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 3) { // 4 pk_mul + 4 pk_add = 8 mul + 8 add unfused
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p0) : "v"(av), "v"(bv));
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p1) : "v"(av), "v"(bv));
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p2) : "v"(av), "v"(bv));
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p3) : "v"(av), "v"(bv));
        } else if (MODE == 4) { // 8 dpp moves
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x0) : "v"(x1));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x1) : "v"(x2));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x2) : "v"(x3));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x3) : "v"(x4));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x4) : "v"(x5));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x5) : "v"(x6));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x6) : "v"(x7));
            asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x7) : "v"(x0));
        } else if (MODE == 5) { // 8 cvt_pk_u8
            unsigned u = __float_as_uint(x0);
            asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0\n v_cvt_pk_u8_f32 %0, %2, 1, %0\n v_cvt_pk_u8_f32 %0, %3, 2, %0\n v_cvt_pk_u8_f32 %0, %4, 3, %0\n"
                         "v_cvt_pk_u8_f32 %0, %1, 0, %0\n v_cvt_pk_u8_f32 %0, %2, 1, %0\n v_cvt_pk_u8_f32 %0, %3, 2, %0\n v_cvt_pk_u8_f32 %0, %4, 3, %0"
                         : "+v"(u) : "v"(x1), "v"(x2), "v"(x3), "v"(x4));
            x0 = __uint_as_float(u);
        } else if (MODE == 7) { // 8 v_fma_mix_f32, f16 low-half source
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x0) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x1) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x2) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x3) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x4) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x5) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x6) : "v"(x7), "v"(a));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x3) : "v"(x7), "v"(a));
        } else if (MODE == 8) { // 8 v_fmac_f32 with an SGPR weight (the kernel's real form)
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x0) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x1) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x2) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x3) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x4) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x5) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x6) : "s"(a), "v"(x7));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x2) : "s"(b), "v"(x7));
        } else if (MODE == 9) { // 8 v_cvt_f32_ubyte
            unsigned u = __float_as_uint(x7);
            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x0) : "v"(u));
            asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x1) : "v"(u));
            asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(x2) : "v"(u));
            asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(x3) : "v"(u));
            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x4) : "v"(u));
            asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x5) : "v"(u));
            asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(x6) : "v"(u));
            asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(x7) : "v"(u));
        } else if (MODE == 6) { // 8 ds_bpermute
            int idx = ((threadIdx.x + 1) & 63) * 4;
            x0 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x0)));
            x1 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x1)));
            x2 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x2)));
            x3 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x3)));
            x4 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x4)));
            x5 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x5)));
            x6 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x6)));
            x7 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x7)));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *in, uint4 *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// 1 read : 4 write, the traffic mix of a x2 upscale
__global__ __launch_bounds__(256) void k_expand4(const uint4 *in, uint4 *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint4 v = in[i];
        out[4 * i] = v; out[4 * i + 1] = v; out[4 * i + 2] = v; out[4 * i + 3] = v;
    }
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    // 1. cvt_pk_u8_f32
    {
        std::vector<float> h = {-1.0f, -0.5f, 0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 2.4999f, 3.5f, 126.5f, 127.5f, 254.5f, 254.9f, 255.0f, 255.5f, 256.0f, 1000.0f, 0.49999997f};
        float *d; unsigned *o; CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, h.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d, o, (int)h.size());
        std::vector<unsigned> r(h.size()); CK(hipMemcpy(r.data(), o, h.size() * 4, hipMemcpyDeviceToHost));
        printf("[cvt_pk_u8_f32]");
        for (size_t i = 0; i < h.size(); ++i) printf(" %g->%u", h[i], r[i]);
        printf("\n");
    }
    // 3. dpp
    {
        int *u, *dn; CK(hipMalloc(&u, 256)); CK(hipMalloc(&dn, 256));
        hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, u, dn);
        int hu[64], hd[64]; CK(hipMemcpy(hu, u, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd, dn, 256, hipMemcpyDeviceToHost));
        printf("[dpp wave_shr:1] lane0=%d lane1=%d lane15=%d lane16=%d lane31=%d lane32=%d lane63=%d\n", hu[0], hu[1], hu[15], hu[16], hu[31], hu[32], hu[63]);
        printf("[dpp wave_shl:1] lane0=%d lane15=%d lane16=%d lane31=%d lane32=%d lane62=%d lane63=%d\n", hd[0], hd[15], hd[16], hd[31], hd[32], hd[62], hd[63]);
    }
    // 2. VALU rates: 256 CUs x 8 blocks x 256 threads -> 8 waves/SIMD
    {
        float *o; CK(hipMalloc(&o, 256 * 16 * 256 * 4));
        unsigned long long *clk; CK(hipMalloc(&clk, 16));
        const int iters = 20000;
        const double lane_ops = 8.0 * iters * 256.0 * 2048;
        const char *names[] = {"v_fma_f32 x8", "v_pk_fma_f32 x4 (=8 fma)", "v_mul+v_add x8", "v_pk_mul+v_pk_add x4", "v_mov_dpp wave_sh x8",
                               "v_cvt_pk_u8_f32 x8", "ds_bpermute x8", "v_fma_mix_f32 (f16 src) x8", "v_fmac_f32 sgpr-weight x8", "v_cvt_f32_ubyteN x8"};
        for (int occ = 0; occ < 3; ++occ) {
            const int blocks = occ == 0 ? 2048 : (occ == 1 ? 1024 : 512); // 8 / 4 / 2 waves per SIMD
            for (int m = 0; m < 10; ++m) {
                float ms = 0;
                auto run = [&](auto kern) { ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, 1.0001f, 0.5f, iters, clk); }, 3); };
                switch (m) { case 0: run(k_valu<0>); break; case 1: run(k_valu<1>); break; case 2: run(k_valu<2>); break; case 3: run(k_valu<3>); break;
                             case 4: run(k_valu<4>); break; case 5: run(k_valu<5>); break; case 6: run(k_valu<6>); break; case 7: run(k_valu<7>); break;
                             case 8: run(k_valu<8>); break; case 9: run(k_valu<9>); break; }
                unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
                const double ops = lane_ops * blocks / 2048.0;
                const double ghz = hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0;
                const double cyc_per_instr = (double)hc[0] / (8.0 * iters) / (blocks / 256.0); // per SIMD: waves/SIMD = blocks/256
                printf("[valu %d waves/SIMD] %-28s %8.3f ms  %7.2f T lane-instr/s  clock %.2f GHz  %.2f cyc per wave-instr per SIMD\n",
                       blocks / 256, names[m], ms, ops / ms / 1e9, ghz, cyc_per_instr);
            }
        }
    }
    // 4. copy ceilings
    {
        const size_t bytes = (size_t)2 << 30; // 2 GiB src, beyond the 256 MiB Infinity Cache
        uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes * 4)); CK(hipMemset(a, 1, bytes));
        const size_t n = bytes / 16;
        for (int blocks : {2048, 4096, 16384}) {
            float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
            printf("[copy 2GiB grid %5d] %.3f ms  %.2f TB/s (read+write)\n", blocks, ms, 2.0 * bytes / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_expand4, dim3(blocks), dim3(256), 0, 0, a, b, n / 2); }, 5);
            printf("[expand 1R:4W grid %5d] %.3f ms  %.2f TB/s (read+write)\n", blocks, ms, 5.0 * (bytes / 2) / ms / 1e9);
        }
        float ms = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("[hipMemcpyDtoD 2GiB] %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}

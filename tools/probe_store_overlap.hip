// probe_store_overlap.hip -- how well do a wave's VALU work and its streaming stores overlap on gfx950?
// The x2 resize kernel is a two-resource loop: per output row a wave issues a few hundred VALU instructions
// and two 1-KiB stores, 12 waves per CU.  Its arithmetic alone and its stores alone each take about the same
// time, and together they take almost the sum.  This probe reproduces that shape with nothing else in it --
// F dependent-chain FMAs per "phase" (8 independent accumulators) and 2 x 16-B-per-lane stores per phase into
// a 4K-frame-shaped buffer -- and varies WHERE the stores sit and how they are issued.
//   usage: probe_store_overlap [frames=64]
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_store_overlap.hip -o tools/probe_store_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = 3840 * 4, kRows = 2160, kStrips = 7, kRowsPerWave = 72; // 30 row blocks x 7 strips per frame

enum Pattern { VALU_ONLY = 0, STORES_ONLY, END_OF_PHASE, SPREAD_HALVES, PAIR_OF_PHASES, GLOBAL_STORES, SPREAD_SETPRIO, NT_STORES, SPREAD_QUARTERS_B64,
               LOADS_ONLY, LOADS_STORES, LOADS_STORES_VALU, CACHED_LOADS_STORES, LOADS_FIRST_STORES, LOADS_STORES_D4 };

#define FMA8() \
    asm volatile("v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\t" \
                 "v_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9\n\tv_fmac_f32 %6, %8, %9\n\tv_fmac_f32 %7, %8, %9" \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(va), "v"(vb))

template <int P>
__global__ __launch_bounds__(256) void k_probe(unsigned char *out, const unsigned char *in, int f8 /* FMA8 groups per phase */, float a, float b, unsigned nwaves)
{
    extern __shared__ unsigned char pad[]; // occupancy control only: 3 blocks (12 waves) per CU like the real kernel
    const int lane = threadIdx.x & 63;
    const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (gw >= nwaves) return;
    const unsigned strip = gw % kStrips, rb = (gw / kStrips) % (kRows / kRowsPerWave), frame = gw / (kStrips * (kRows / kRowsPerWave));
    const size_t frame_bytes = (size_t)kRowBytes * kRows;
    unsigned char *fbase = out + (size_t)frame * frame_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(fbase, 0, (unsigned)frame_bytes, 0x00020000);
    unsigned off = (rb * kRowsPerWave) * kRowBytes + strip * 2048 + lane * 32;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    float va = a + lane * 0.0f, vb = b;
    asm volatile("" : "+v"(va), "+v"(vb));
    if (P == SPREAD_SETPRIO) __builtin_amdgcn_s_setprio(0);
    if (P >= LOADS_ONLY) {
        // one 1-KiB input row per wave per STEP (2 phases = 2 output rows), requested D steps ahead into registers, like the
        // x2 kernel: input frame 1920 x 1080 x 4 B, the wave's strip is 1 KiB of each 7680-B row
        constexpr int D = P == LOADS_STORES_D4 ? 4 : 2;
        const unsigned char *ibase = in + (size_t)frame * (7680u * 1080u);
        const unsigned ioff0 = (rb * (kRowsPerWave / 2)) * 7680u + strip * 1024 + lane * 16;
        auto row_ptr = [&](int step) {
            int r = step < kRowsPerWave / 2 ? step : kRowsPerWave / 2 - 1;
            if (P == CACHED_LOADS_STORES) return in + (unsigned)(r & 15) * 7680u + strip * 1024 + lane * 16;
            return ibase + ioff0 + (unsigned)r * 7680u;
        };
        u32x4 raw[D];
#pragma unroll
        for (int j = 0; j < D; ++j) raw[j] = *reinterpret_cast<const u32x4 *>(row_ptr(j));
        for (int step = 0; step < kRowsPerWave / 2; step += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int st = step + j;
                u32x4 lo, hi;
                if (P == LOADS_FIRST_STORES) { // consume + re-request BEFORE this step's stores
                    x[0] += __uint_as_float(raw[j].x & 0x3fffffffu);
                    raw[j] = *reinterpret_cast<const u32x4 *>(row_ptr(st + D));
                }
                for (int ph = 0; ph < 2; ++ph) {
                    if (P == LOADS_STORES_VALU) for (int i = 0; i < f8; ++i) FMA8();
                    if (P != LOADS_ONLY) {
                        lo = u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
                        hi = u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
                        __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, 0);
                    }
                    off += kRowBytes;
                    if (ph == 0 && P != LOADS_FIRST_STORES) { // between the two phases, as the x2 kernel does
                        x[0] += __uint_as_float(raw[j].x & 0x3fffffffu);
                        raw[j] = *reinterpret_cast<const u32x4 *>(row_ptr(st + D));
                    }
                }
            }
        }
        float sacc = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
        if (sacc == 12345.678f) out[gw] = 1;
        return;
    }
    for (int row = 0; row < kRowsPerWave; ++row, off += kRowBytes) {
        u32x4 lo, hi;
        auto pack = [&]() {
            lo = u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
            hi = u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
        };
        if (P == VALU_ONLY) {
            for (int i = 0; i < f8; ++i) FMA8();
        } else if (P == STORES_ONLY) {
            pack();
            __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, 0);
        } else if (P == END_OF_PHASE || P == NT_STORES) {
            for (int i = 0; i < f8; ++i) FMA8();
            pack();
            __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, P == NT_STORES ? 2 : 0);
            __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, P == NT_STORES ? 2 : 0);
        } else if (P == GLOBAL_STORES) {
            for (int i = 0; i < f8; ++i) FMA8();
            pack();
            *reinterpret_cast<u32x4 *>(fbase + off) = lo;
            *reinterpret_cast<u32x4 *>(fbase + off + 16) = hi;
        } else if (P == SPREAD_HALVES || P == SPREAD_SETPRIO) {
            for (int i = 0; i < f8 / 2; ++i) FMA8();
            pack();
            __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 0);
            for (int i = f8 / 2; i < f8; ++i) FMA8();
            pack();
            __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, 0);
        } else if (P == SPREAD_QUARTERS_B64) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            // (not the same bytes per lane per store: 8 B per lane, 512 B per wave instruction, 4 stores per phase)
            for (int q = 0; q < 4; ++q) {
                for (int i = 0; i < f8 / 4; ++i) FMA8();
                u32x2 v = {__float_as_uint(x[2 * q]), __float_as_uint(x[2 * q + 1])};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, off - lane * 32 + q * 512 + lane * 8, 0, 0);
            }
        } else if (P == PAIR_OF_PHASES) { // burstier: the four stores of two phases together
            for (int i = 0; i < f8; ++i) FMA8();
            if (row & 1) {
                pack();
                __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off - kRowBytes, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off - kRowBytes + 16, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, 0);
            }
        }
    }
    if (P == VALU_ONLY) { // keep the arithmetic alive
        float s = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
        if (s == 12345.678f) out[gw] = 1;
    }
}

template <int P>
static int run(const char *name, unsigned char *out, const unsigned char *in, int frames, int f8, int lds_bytes, std::vector<float> *res)
{
    const unsigned nwaves = (unsigned)frames * kStrips * (kRows / kRowsPerWave);
    const dim3 grid((nwaves + 3) / 4), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_probe<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    std::vector<float> t;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<P>, grid, block, lds_bytes, 0, out, in, f8, 1.0001f, 0.5f, nwaves);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) t.push_back(ms * 1e3f / frames);
    }
    std::sort(t.begin(), t.end());
    res->push_back(t[t.size() / 2]);
    (void)name;
    return 0;
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 64;
    unsigned char *out;
    CK(hipMalloc(&out, (size_t)frames * kRowBytes * kRows));
    unsigned char *in;
    CK(hipMalloc(&in, (size_t)frames * 7680 * 1080));
    CK(hipMemset(in, 1, (size_t)frames * 7680 * 1080));
    const char *names[] = {"VALU only", "stores only", "2 stores at end of phase", "1 store per half phase", "4 stores per 2 phases",
                           "global_store (end of phase)", "half phase + setprio0", "nt stores (end of phase)", "4 x 8-B stores per phase",
                           "loads only (1 KiB / step)", "loads + stores", "loads + stores + VALU", "cached loads + stores", "loads before stores", "loads (4 ahead) + stores"};
    printf("frames=%d; per 4K frame: %d wave-phases, 2 x 1 KiB stores each (%.1f MB); us per frame, median of 3\n", frames,
           kStrips * kRows, kStrips * kRows * 2048 / 1e6);
    for (int waves_per_simd : {3, 8}) {
        const int blocks_per_cu = waves_per_simd; // 4 waves per block, one per SIMD
        const int lds = blocks_per_cu >= 8 ? 0 : (160 * 1024 / blocks_per_cu) - 1024;
        printf("-- %d waves per SIMD (LDS pad %d B per block)\n", waves_per_simd, lds);
        printf("%-32s", "FMAs per phase:");
        const int f8s[] = {15, 30, 45, 60};
        for (int f8 : f8s) printf("%8d", f8 * 8);
        printf("\n");
        for (int p = 0; p < 15; ++p) {
            std::vector<float> res;
            for (int f8 : f8s) {
                int rc = 0;
                switch (p) {
                case 0: rc = run<VALU_ONLY>(names[p], out, in, frames, f8, lds, &res); break;
                case 1: rc = run<STORES_ONLY>(names[p], out, in, frames, f8, lds, &res); break;
                case 2: rc = run<END_OF_PHASE>(names[p], out, in, frames, f8, lds, &res); break;
                case 3: rc = run<SPREAD_HALVES>(names[p], out, in, frames, f8, lds, &res); break;
                case 4: rc = run<PAIR_OF_PHASES>(names[p], out, in, frames, f8, lds, &res); break;
                case 5: rc = run<GLOBAL_STORES>(names[p], out, in, frames, f8, lds, &res); break;
                case 6: rc = run<SPREAD_SETPRIO>(names[p], out, in, frames, f8, lds, &res); break;
                case 7: rc = run<NT_STORES>(names[p], out, in, frames, f8, lds, &res); break;
                case 8: rc = run<SPREAD_QUARTERS_B64>(names[p], out, in, frames, f8, lds, &res); break;
                case 9: rc = run<LOADS_ONLY>(names[p], out, in, frames, f8, lds, &res); break;
                case 10: rc = run<LOADS_STORES>(names[p], out, in, frames, f8, lds, &res); break;
                case 11: rc = run<LOADS_STORES_VALU>(names[p], out, in, frames, f8, lds, &res); break;
                case 12: rc = run<CACHED_LOADS_STORES>(names[p], out, in, frames, f8, lds, &res); break;
                case 13: rc = run<LOADS_FIRST_STORES>(names[p], out, in, frames, f8, lds, &res); break;
                case 14: rc = run<LOADS_STORES_D4>(names[p], out, in, frames, f8, lds, &res); break;
                }
                if (rc) return rc;
            }
            printf("%-32s", names[p]);
            for (float v : res) printf("%8.2f", v);
            printf("\n");
            fflush(stdout);
        }
    }
    CK(hipFree(out));
    return 0;
}

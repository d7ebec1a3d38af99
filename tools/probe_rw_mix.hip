// probe_rw_mix.hip -- HBM reads mixed with HBM writes in the access shape of the x2 resize kernel (gfx950).
// tools/probe_store_overlap.hip found: a wave that reads one 1-KiB input row per step and writes two 2-KiB
// output rows per step, 12 waves per CU, each wave walking down its own 36-row block, moves 1080p -> 4K
// frames at 9.8 us / frame (4.0 TB/s) although the loads alone take 1.3 us and the stores alone 5.5 us --
// at any occupancy and any prefetch distance, with or without arithmetic; with the loads served from L2 it is
// 5.8 us.  This probe varies what the memory system sees: rows per wave (how many independent row streams are
// open at once), the store shape (16-B pieces at 32-B stride vs contiguous 1 KiB per instruction), the cache
// policy of the loads, and a "touch" pass that brings a chunk of input frames into the Infinity Cache first.
//   usage: probe_rw_mix [frames=96]
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_rw_mix.hip -o tools/probe_rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kInRow = 1920 * 4, kInRows = 1080, kOutRow = 3840 * 4, kStrips = 7;

// STEPS input rows per wave (2 output rows each); CONTIG: each store instruction writes 1 KiB contiguous;
// AUX: cache policy bits of the loads (0 default, 2 nt, 1 sc0 ...); MODE 0 loads+stores, 1 stores only, 2 loads only
template <int STEPS, bool CONTIG, int AUX, int MODE>
__global__ __launch_bounds__(256) void k_mix(unsigned char *out, const unsigned char *in, unsigned nwaves, unsigned first_frame)
{
    extern __shared__ unsigned char pad[];
    const int lane = threadIdx.x & 63;
    const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (gw >= nwaves) return;
    constexpr unsigned nrb = kInRows / STEPS;
    const unsigned strip = gw % kStrips, rb = (gw / kStrips) % nrb, frame = first_frame + gw / (kStrips * nrb);
    const size_t out_frame = (size_t)kOutRow * kInRows * 2, in_frame = (size_t)kInRow * kInRows;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)frame * out_frame, 0, (unsigned)out_frame, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(in) + (size_t)frame * in_frame, 0,
                                                                         (unsigned)in_frame, 0x00020000);
    unsigned off = (rb * STEPS * 2) * kOutRow + strip * 2048 + (CONTIG ? lane * 16 : lane * 32);
    unsigned ioff = (rb * STEPS) * kInRow + strip * 1024 + lane * 16;
    constexpr int D = STEPS >= 2 ? 2 : 1;
    u32x4 raw[D];
    unsigned acc = lane;
    if (MODE != 1) {
#pragma unroll
        for (int j = 0; j < D; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ioff + (unsigned)j * kInRow, 0, AUX);
    }
    for (int step = 0; step < STEPS; step += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int st = step + j;
            for (int ph = 0; ph < 2; ++ph) {
                if (MODE != 2) {
                    const u32x4 lo = {acc, acc + 1, acc + 2, acc + 3}, hi = {acc + 4, acc + 5, acc + 6, acc + 7};
                    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + (CONTIG ? 1024 : 16), 0, 0);
                }
                off += kOutRow;
                if (ph == 0 && MODE != 1) {
                    acc += raw[j].x & 0xffu;
                    int nxt = st + D;
                    nxt = nxt < STEPS ? nxt : STEPS - 1;
                    raw[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ioff + (unsigned)nxt * kInRow, 0, AUX);
                }
            }
        }
    }
    if (acc == 0x12345678u) out[gw] = 1;
}

// read every byte of frames [f0, f0 + n) once, 16 B per lane, result discarded: brings them into the memory-side cache
__global__ __launch_bounds__(256) void k_touch(const unsigned char *in, size_t bytes, unsigned *sink)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= bytes) return;
    const u32x4 v = *reinterpret_cast<const u32x4 *>(in + i);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { hipEventCreate(&e0); hipEventCreate(&e1); }
    void start() { hipEventRecord(e0); }
    float stop_us() { hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f; }
};

template <int STEPS, bool CONTIG, int AUX, int MODE>
static float run(unsigned char *out, const unsigned char *in, int frames, int waves_per_simd, int chunk = 0, unsigned *sink = nullptr)
{
    const int lds = waves_per_simd >= 8 ? 0 : (160 * 1024 / waves_per_simd) - 1024;
    hipFuncSetAttribute((const void *)k_mix<STEPS, CONTIG, AUX, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const unsigned per_frame = kStrips * (kInRows / STEPS);
    Timer t;
    std::vector<float> v;
    for (int rep = 0; rep < 4; ++rep) {
        t.start();
        if (chunk == 0) {
            const unsigned nw = (unsigned)frames * per_frame;
            hipLaunchKernelGGL((k_mix<STEPS, CONTIG, AUX, MODE>), dim3((nw + 3) / 4), dim3(256), lds, 0, out, in, nw, 0u);
        } else {
            for (int f = 0; f < frames; f += chunk) {
                const int n = std::min(chunk, frames - f);
                const size_t bytes = (size_t)n * kInRow * kInRows;
                hipLaunchKernelGGL(k_touch, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, in + (size_t)f * kInRow * kInRows, bytes, sink);
                const unsigned nw = (unsigned)n * per_frame;
                hipLaunchKernelGGL((k_mix<STEPS, CONTIG, AUX, MODE>), dim3((nw + 3) / 4), dim3(256), lds, 0, out, in, nw, (unsigned)f);
            }
        }
        const float us = t.stop_us();
        if (rep) v.push_back(us / frames);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 96;
    unsigned char *out, *in;
    unsigned *sink;
    CK(hipMalloc(&out, (size_t)frames * kOutRow * kInRows * 2));
    CK(hipMalloc(&in, (size_t)frames * kInRow * kInRows));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, (size_t)frames * kInRow * kInRows));
    printf("frames=%d, 1080p -> 4K shape: per frame 7.7 MB read (7 strips of 1 KiB) + 31 MB written; us per frame, median of 3\n", frames);
    for (int w : {3, 8}) {
        printf("-- %d waves per SIMD\n", w);
        printf("%-64s %8s %8s %8s\n", "input rows per wave:", "36", "12", "4");
#define ROW3(LABEL, C, A, M) printf("%-64s %8.2f %8.2f %8.2f\n", LABEL, run<36, C, A, M>(out, in, frames, w), run<12, C, A, M>(out, in, frames, w), run<4, C, A, M>(out, in, frames, w)); fflush(stdout)
        ROW3("stores only, 16-B pieces at 32-B stride", false, 0, 1);
        ROW3("stores only, 1 KiB contiguous per instruction", true, 0, 1);
        ROW3("loads only", false, 0, 2);
        ROW3("loads + stores (strided pieces)", false, 0, 0);
        ROW3("loads + stores (contiguous)", true, 0, 0);
        ROW3("nt loads + stores (contiguous)", true, 2, 0);
        ROW3("sc1 loads + stores (contiguous)", true, 16, 0);
        ROW3("sc0 sc1 loads + stores (contiguous)", true, 17, 0);
        printf("%-64s %8.2f %8.2f %8.2f\n", "1 input row per wave (1 load, 4 stores; raster order), strided / contiguous / -",
               run<1, false, 0, 0>(out, in, frames, w), run<1, true, 0, 0>(out, in, frames, w), 0.0f);
        for (int chunk : {2, 4, 6}) {
            printf("touch %d frames, then loads + stores (contiguous) on them         %8.2f %8.2f %8.2f\n", chunk,
                   run<36, true, 0, 0>(out, in, frames, w, chunk, sink), run<12, true, 0, 0>(out, in, frames, w, chunk, sink),
                   run<4, true, 0, 0>(out, in, frames, w, chunk, sink));
            fflush(stdout);
        }
    }
    return 0;
}

// probe_row_store_shapes.hip -- round 6: can a row-walking x2 kernel store its output rows WITHOUT the turn through LDS?
// The x2 resize kernel's lane computes 32 contiguous bytes of an output row (two 16-byte pieces A, B at 32 * lane).  Stored as they
// are (two instructions of 16-byte pieces at a 32-byte lane stride) the stream loses a third of its rate once HBM reads share it
// (tools/probe_rw_mix.hip, round 2), so the product turns every row round in LDS: each store instruction then writes one
// contiguous KiB in lane order.  gfx950's v_permlane32_swap_b32 offers a third shape with no LDS: swap(A, B) leaves, in A, the
// pieces of lanes 0..31 (A of lane l at 32 l in lanes 0..31, B of lane l - 32 at 32 (l - 32) + 16 in lanes 32..63) -- ONE
// contiguous KiB per instruction again, but in an interleaved lane order.  This probe times the three shapes in the access
// pattern of probe_rw_mix (a wave reads one 1-KiB input row and writes two 2-KiB output rows per step, 36 rows per wave), with
// default and non-temporal stores.    usage: probe_row_store_shapes [frames=96]
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_row_store_shapes.hip -o tools/probe_row_store_shapes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kInRow = 1920 * 4, kInRows = 1080, kOutRow = 3840 * 4, kStrips = 7;

// SHAPE 0: the lane's own pieces (16 B at a 32-B stride, twice); 1: linear KiB per instruction (what the LDS turn produces; here
// the values are simply generated in that layout: the turn's cost is NOT in this number); 2: permlane32_swap, interleaved KiB
template <int STEPS, int SHAPE, int SAUX, bool SWAP_REAL>
__global__ __launch_bounds__(256) void k_rows(unsigned char *out, const unsigned char *in, unsigned nwaves)
{
    extern __shared__ unsigned char pad[];
    const int lane = threadIdx.x & 63;
    const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (gw >= nwaves) return;
    constexpr unsigned nrb = kInRows / STEPS;
    const unsigned strip = gw % kStrips, rb = (gw / kStrips) % nrb, frame = gw / (kStrips * nrb);
    const size_t out_frame = (size_t)kOutRow * kInRows * 2, in_frame = (size_t)kInRow * kInRows;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)frame * out_frame, 0, (unsigned)out_frame, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(in) + (size_t)frame * in_frame, 0,
                                                                         (unsigned)in_frame, 0x00020000);
    const unsigned lane_off = SHAPE == 0 ? lane * 32 : (SHAPE == 1 ? lane * 16 : (lane < 32 ? lane * 32 : (lane - 32) * 32 + 16));
    const unsigned second = SHAPE == 0 ? 16 : 1024;
    unsigned off = (rb * STEPS * 2) * kOutRow + strip * 2048 + lane_off;
    unsigned ioff = (rb * STEPS) * kInRow + strip * 1024 + lane * 16;
    constexpr int D = 2;
    u32x4 raw[D];
    unsigned acc = lane;
#pragma unroll
    for (int j = 0; j < D; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ioff + (unsigned)j * kInRow, 0, 0);
    for (int step = 0; step < STEPS; step += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int st = step + j;
            for (int ph = 0; ph < 2; ++ph) {
                u32x4 lo = {acc, acc + 1, acc + 2, acc + 3}, hi = {acc + 4, acc + 5, acc + 6, acc + 7};
                if (SHAPE == 2 && SWAP_REAL) { // the four swaps the real kernel would issue
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x2 r = __builtin_amdgcn_permlane32_swap(lo[k], hi[k], false, false);
                        lo[k] = r.x, hi[k] = r.y;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, SAUX);
                __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + second, 0, SAUX);
                off += kOutRow;
                if (ph == 0) {
                    acc += raw[j].x & 0xffu;
                    int nxt = st + D;
                    nxt = nxt < STEPS ? nxt : STEPS - 1;
                    raw[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ioff + (unsigned)nxt * kInRow, 0, 0);
                }
            }
        }
    }
    if (acc == 0x12345678u) out[gw] = 1;
}

template <int SHAPE, int SAUX, bool SWAP_REAL>
static float run(unsigned char *out, const unsigned char *in, int frames)
{
    constexpr int STEPS = 36;
    const int lds = (160 * 1024 / 3) - 1024; // three waves per SIMD, as the x2 kernel has
    hipFuncSetAttribute((const void *)k_rows<STEPS, SHAPE, SAUX, SWAP_REAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const unsigned nw = (unsigned)frames * kStrips * (kInRows / STEPS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> v;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rows<STEPS, SHAPE, SAUX, SWAP_REAL>), dim3((nw + 3) / 4), dim3(256), lds, 0, out, in, nw);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) v.push_back(ms * 1e3f / frames);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 96;
    unsigned char *out, *in;
    CK(hipMalloc(&out, (size_t)frames * kOutRow * kInRows * 2));
    CK(hipMalloc(&in, (size_t)frames * kInRow * kInRows));
    CK(hipMemset(in, 1, (size_t)frames * kInRow * kInRows));
    printf("frames=%d, 1080p -> 4K shape, 36 input rows per wave, 3 waves per SIMD: us per frame (median of 5), default / nt stores\n", frames);
    for (int round = 0; round < 2; ++round) {
        printf("16-B pieces at a 32-B lane stride (the lane's own bytes)          %7.2f %7.2f\n", run<0, 0, false>(out, in, frames), run<0, 2, false>(out, in, frames));
        printf("one KiB per instruction, lane order (after the LDS turn)          %7.2f %7.2f\n", run<1, 0, false>(out, in, frames), run<1, 2, false>(out, in, frames));
        printf("one KiB per instruction, permlane32_swap order (addresses only)   %7.2f %7.2f\n", run<2, 0, false>(out, in, frames), run<2, 2, false>(out, in, frames));
        printf("one KiB per instruction, permlane32_swap order, with the 4 swaps  %7.2f %7.2f\n", run<2, 0, true>(out, in, frames), run<2, 2, true>(out, in, frames));
        fflush(stdout);
    }
    return 0;
}

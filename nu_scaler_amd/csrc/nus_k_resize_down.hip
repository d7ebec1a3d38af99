// nus_k_resize_down.hip -- separable resize for vertical DOWN-scaling (tap windows of up to 31 rows): the
// filters of image 0.24.9's imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251 and
// Nu_scale/src/capture/common.rs:56, :359-360 (captured frames are resized to the target with them).
#include "nus_device.hpp"
#include "nus_kernels.hpp"

namespace nus {

namespace {

#ifndef NUS_DOWN_DEPTH
#define NUS_DOWN_DEPTH 4 // input rows in flight per wave through the LDS-DMA ring (power of two); 0 = round 2's form: the next
                         // row in registers (dev macro, A/B timing)
#endif
#ifndef NUS_DOWN_RING_MAX_VC
#define NUS_DOWN_RING_MAX_VC 2 // the ring serves the shapes with up to this many columns per lane (ratios up to ~x1.8 down);
                               // wider footprints keep the next row in registers: measured per shape in
                               // profiles/r03_resize_down_lds_dma_ring_ab.txt (1080p -> 720p 7.7 -> 6.9 us, 1440p -> 1080p 17.2 ->
                               // 16.2; 4K -> 1080p 22.5 -> 23.0, 4K -> 720p 20.6 -> 22.0: those are bound by their arithmetic and
                               // their LDS passes, not by the rows in flight)
#endif
constexpr int kDownDepth = NUS_DOWN_DEPTH;
constexpr int down_depth(int vc) { return vc <= NUS_DOWN_RING_MAX_VC ? kDownDepth : 0; }
static_assert(kDownDepth == 0 || (kDownDepth & (kDownDepth - 1)) == 0, "ring depth must be a power of two");
constexpr uint32_t kDownSlack = 32; // zeroed LDS entries behind the row: the fixed-length horizontal loop reads up to HT - 1 past a window

// Streaming form of vertical_sample -> horizontal_sample for ratio >= 1.  On a down-scale every input
// row feeds up to 7 output rows (window 6*ratio+1 rows, outputs ratio rows apart), so instead of
// gathering a 13..31-row window per output row, a wave walks DOWN THE INPUT ROWS ONCE and keeps the
// vertical sums of the output rows in flight in registers:
//   * lane l holds VC input columns (l, l+64, ..) of the wave's 64-output segment footprint;
//   * 7 accumulator slots, output row y lives in slot y % 7.  The host lays the vertical weights out
//     PER INPUT ROW (build_down_stream_tables): 7 weights -- the row's weight in each slot, 0 where it
//     is outside that slot's window -- and the output row, if any, whose window ends on it.  One
//     32-byte scalar load per input row replaces all per-slot window arithmetic; every slot takes
//     acc[s] += w[s] * row.  Taps are added in increasing row order, exactly the order of the
//     per-output loop, and an added +-0 changes nothing: bit-identical in EXACT mode (mul + add);
//   * the row that completes a window writes the slot's sums to the wave's LDS row; each lane then
//     sums its output's horizontal taps from LDS (weights in registers, or in LDS when the accumulators
//     leave no room for them: 3+ columns per lane), packs and stores, and the
//     slot is cleared.  Windows cut by the bottom border end together on the last input row: the
//     table gives each of them a pseudo-row of zero weights behind it, so the loop never sees two.
// A block of output rows starts at its first window's first row with cleared slots; windows of the
// previous block that are still open there complete with partial sums and are dropped (y < y_begin).
// Every input byte is read once per row block (+ that window fill), every vertical product is
// computed once, and the host has checked that no slot ever holds two open windows.
constexpr int kSlots = 7;          // == nus::kDownSlots (nus_tables.hpp)
constexpr uint32_t kNone = 0xFFFFFFFFu;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in this kernel uses it
// Request 4 B per lane from `base + off` (base wave-uniform, off < 4 GiB) into the 256-byte LDS piece at byte offset `lds`
// (wave-uniform): lane l lands at lds + 4 l.  No VGPR destination, invisible to the compiler's s_waitcnt insertion.
__device__ __forceinline__ void dma_row4(const void *base, uint32_t off, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
// Wait until at most N of this wave's vector memory instructions are outstanding (they retire in issue order).
// BACK (for tools/check_hidden_loads.py): the wait is for the BACK-th most recent row piece.
template <int N, int BACK>
__device__ __forceinline__ void down_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}
// The reads of a ring slot must have RETURNED before the slot is requested again: nothing orders a queued ds_read behind a
// later LDS-DMA write, and with 16 waves per CU running their horizontal passes out of LDS the read can sit in the LDS queue
// longer than a request that hits in L1 takes to land (seen as a wrong input row for some lanes of a wave, a few hundred
// pixels per 4K -> 1080p batch, never on a single frame).
__device__ __forceinline__ void down_reads_done()
{
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
}

template <bool EXACT, int VC, int HT>
__global__ __launch_bounds__(256) void k_resize_down(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const int32_t *__restrict__ lxt, const uint32_t *__restrict__ nxt, const float *__restrict__ wxt,
    const int32_t *__restrict__ lyt, const uint32_t *__restrict__ rowtab, const int32_t *__restrict__ done_row,
    uint32_t stride, uint32_t iw, uint32_t ih, uint32_t ow, uint32_t oh, uint32_t rows_per_block, uint32_t ncols_max,
    size_t in_frame_px, size_t out_frame_px, uint32_t sel, uint32_t seg_w)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = kSlots;
    // per wave: the V row (ncols_max + slack float4), then the lanes' horizontal weights [HT][64] (registers are
    // what limits the waves per SIMD here; a tap's weight is one conflict-free ds_read_b32 away)
    constexpr int D = down_depth(VC);
    // the four waves' row rings first (LDS-DMA takes its LDS address from M0: keep it a small offset), then per wave the rest
    constexpr size_t ring_floats = (size_t)D * VC * kWave;
    const size_t wave_floats = (size_t)(ncols_max + kDownSlack) * 4 + (size_t)HT * kWave;
    float *const wave_lds = reinterpret_cast<float *>(smem) + 4 * ring_floats + (size_t)threadIdx.y * wave_floats;
    float4 *s_v = reinterpret_cast<float4 *>(wave_lds);
    float *s_hw = wave_lds + (size_t)(ncols_max + kDownSlack) * 4 + threadIdx.x;
    // row ring: D slots of VC pieces of 64 pixels (lane l's column m of the row in slot k at [(k VC + m) 64 + l])
    uint32_t *s_ring = reinterpret_cast<uint32_t *>(smem) + (size_t)threadIdx.y * ring_floats;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_ring);
    const GridPos g = xcd_contiguous_pos(); // row blocks that share their window-fill rows behind one L2
    const uint32_t seg = __builtin_amdgcn_readfirstlane(g.x * 4 + threadIdx.y);
    // a wave's segment is seg_w <= 64 output columns: the host picks the width whose footprint fills whole columns-per-lane
    // (2x down: 58 outputs over 128 input columns, VC = 2, instead of 64 over 140, VC = 3 with a third of the lanes' vertical
    // work wasted)
    const uint32_t X0 = seg * seg_w;
    if (X0 >= ow) return; // whole wave; no workgroup barriers below
    const uint32_t Xlast = umin(X0 + seg_w, ow) - 1;
    const int32_t cmin = lxt[X0];
    const int32_t ncols = lxt[Xlast] + (int32_t)nxt[Xlast] - cmin; // <= 64 * VC (host-checked)
    if (threadIdx.x < kDownSlack) s_v[ncols + threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t x = X0 + threadIdx.x;
    const bool lane_active = threadIdx.x < seg_w && x < ow;
    const uint32_t xo = lane_active ? x : X0;
    const int32_t hl = lxt[xo] - cmin;
    // the lane's horizontal weights, zero beyond its window (HT >= the widest window, host-checked); adding
    // +-0 leaves every sum as it is, so the fixed trip count does not change a bit
    // With 3+ columns per lane the accumulators leave no room for HT weight registers next to the H pass's LDS reads
    // (180 VGPRs, 2 waves per SIMD, and the kernel waits on HBM): the weights then live in LDS and the tap loop is rolled
    // (128 VGPRs, 4 waves).  With 1-2 columns the registers are there and the unrolled loop is faster.
    constexpr bool HW_LDS = VC >= 3;
    float hw[HW_LDS ? 1 : HT];
    {
        const uint32_t hn = nxt[xo];
        const float *wx = wxt + (size_t)xo * stride;
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            const float wk = (uint32_t)k < hn ? wx[(uint32_t)k < stride ? k : 0] : 0.0f;
            if (HW_LDS) s_hw[k * kWave] = wk;
            else hw[k] = wk;
        }
    }
    const uint32_t y_begin = g.y * rows_per_block;
    const uint32_t y_end = umin(y_begin + rows_per_block, oh);
    const uint32_t *base = in + (size_t)g.z * in_frame_px;
    uint32_t *dst = out + (size_t)g.z * out_frame_px + x;

    // this lane's VC input columns: lane, lane + 64, ...: each load of the wave is one contiguous 256 B and
    // its LDS writes are 64 consecutive float4; clamped into the row (columns past the footprint are never read back)
    uint32_t col[VC];
#pragma unroll
    for (int m = 0; m < VC; ++m) col[m] = umin((uint32_t)cmin + threadIdx.x + kWave * m, iw - 1);
    auto load_row = [&](int32_t r, uint32_t (&raw)[VC]) {
        const uint32_t *row = base + (size_t)umin((uint32_t)r, ih - 1) * iw; // pseudo-rows: any finite pixels
#pragma unroll
        for (int m = 0; m < VC; ++m) raw[m] = row[col[m]];
    };
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    auto load_tab = [&](int32_t r) { return *reinterpret_cast<const u32x8 *>(rowtab + (size_t)r * 8); }; // wave-uniform

    float acc[NS][VC * 4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < VC * 4; ++k) acc[s][k] = 0.0f;
    const int32_t r_first = lyt[y_begin];
    const int32_t r_last = done_row[y_end - 1];
    // Row prefetch.  With up to NUS_DOWN_RING_MAX_VC columns per lane the rows come through a per-wave LDS-DMA ring as in
    // k_lanczos3_x2 (D > 0): row r + D is requested (VC pieces of 256 B, no VGPR destination) once row r has been read out of
    // its slot AND the reads have returned (down_reads_done), and before the read the wave waits with s_waitcnt
    // vmcnt((D - 1) VC): everything but the requests of the D - 1 younger rows has landed.  The stores of completed output
    // rows sit in the same in-order count; not counting them makes the wait stricter (it may also cover a younger row's
    // request), never looser.  Wider footprints (D == 0) keep round 2's form, the next row in registers: they are bound by
    // their arithmetic and LDS passes, and the ring bought nothing there (NUS_DOWN_RING_MAX_VC).
    uint32_t col_off[VC]; // byte offset of the lane's column m inside a row
#pragma unroll
    for (int m = 0; m < VC; ++m) col_off[m] = col[m] * 4u;
    auto request_row = [&](int32_t r, uint32_t slot) {
        const uint32_t ro = umin((uint32_t)r, ih - 1) * (iw * 4u); // pseudo-rows: any finite pixels
#pragma unroll
        for (int m = 0; m < VC; ++m) dma_row4(base, ro + col_off[m], ring_lds + (slot * VC + m) * (kWave * 4u));
    };
    uint32_t raw_next[VC];
    if (D > 0) {
        for (int k = 0; k < D; ++k) request_row(r_first + k < r_last ? r_first + k : r_last, (uint32_t)k);
    } else {
        load_row(r_first, raw_next);
    }
    u32x8 tab_next = load_tab(r_first);

    for (int32_t r = r_first; r <= r_last; ++r) {
        if (D > 0) {
            const uint32_t slot = (uint32_t)(r - r_first) & (uint32_t)(D > 0 ? D - 1 : 0);
            down_wait_vmcnt<(D > 0 ? (D - 1) * VC : 0), (D > 0 ? (D - 1) * VC + 1 : 1)>();
#pragma unroll
            for (int m = 0; m < VC; ++m) raw_next[m] = s_ring[(slot * VC + m) * kWave + threadIdx.x];
            down_reads_done();
            request_row(r + D < r_last ? r + D : r_last, slot); // (past the block: its last row again, never read)
        }
        float p[VC * 4];
#pragma unroll
        for (int m = 0; m < VC; ++m) {
            const uint32_t px = swz(raw_next[m], sel);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[m * 4 + c] = ch_f32(px, c);
        }
        const u32x8 tab = tab_next;
        if (r < r_last) { // one row ahead
            if (D == 0) load_row(r + 1, raw_next);
            tab_next = load_tab(r + 1);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float w = __uint_as_float(tab[s]);
            asm volatile("" : "+v"(w)); // VGPR copy: scalar operands halve the VALU issue rate
#pragma unroll
            for (int q = 0; q < VC * 4; ++q) acc[s][q] = mac<EXACT>(acc[s][q], p[q], w);
        }
        const uint32_t comp = tab[7];
        if (comp == kNone) continue; // wave-uniform
        // the completed window -> LDS row; clear the slot
        const uint32_t y = comp & 0x0FFFFFFFu, slot = comp >> 28;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (slot != (uint32_t)s) continue;
#pragma unroll
            for (int m = 0; m < VC; ++m) {
                const int32_t ci = (int32_t)threadIdx.x + kWave * m;
                if (ci < ncols) s_v[ci] = make_float4(acc[s][m * 4], acc[s][m * 4 + 1], acc[s][m * 4 + 2], acc[s][m * 4 + 3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[s][m * 4 + c] = 0.0f;
            }
        }
        if (y < y_begin) continue; // a window of the previous block: its first rows were not seen here
        // A wave only ever reads the LDS row it wrote itself, and the LDS executes one wave's
        // instructions in order: no workgroup barrier, just keep the compiler from reordering.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
        // taps beyond the lane's window: weight 0 times a finite LDS value
        if (HW_LDS) {
#pragma unroll 4
            for (int k = 0; k < HT; ++k) {
                const float4 v = s_v[hl + k];
                const float wk = s_hw[k * kWave];
                h0 = mac<EXACT>(h0, v.x, wk);
                h1 = mac<EXACT>(h1, v.y, wk);
                h2 = mac<EXACT>(h2, v.z, wk);
                h3 = mac<EXACT>(h3, v.w, wk);
            }
        } else {
#pragma unroll
            for (int k = 0; k < HT; ++k) {
                const float4 v = s_v[hl + k];
                h0 = mac<EXACT>(h0, v.x, hw[k]);
                h1 = mac<EXACT>(h1, v.y, hw[k]);
                h2 = mac<EXACT>(h2, v.z, hw[k]);
                h3 = mac<EXACT>(h3, v.w, hw[k]);
            }
        }
        if (lane_active)
            dst[(size_t)y * ow] = pack_u8<EXACT>(h3, 3, pack_u8<EXACT>(h2, 2, pack_u8<EXACT>(h1, 1, pack_u8<EXACT>(h0, 0, 0u))));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next completed row overwrites s_v
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace

hipError_t launch_resize_down(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max, uint32_t max_taps_x,
                              uint32_t seg_w)
{
    if (seg_w < 1 || seg_w > kWave) return hipErrorInvalidValue;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    const uint32_t vc = cdiv(ncols_max, kWave);
    if (vc < 1 || vc > 5 || max_taps_x > 32 || !T.lz_down_rows || !T.lz_down_done) return hipErrorInvalidValue;
    const bool wide = max_taps_x > 16; // horizontal weights per lane: 16 or 32 registers
    const size_t lds = (size_t)4 * ((size_t)(ncols_max + kDownSlack) * sizeof(float4) + (size_t)(wide ? 32 : 16) * kWave * sizeof(float) +
                                    (size_t)down_depth((int)vc) * vc * kWave * sizeof(uint32_t));
    if ((uint64_t)L.iw * L.ih * 4 >= (1ull << 32)) return hipErrorInvalidValue; // 32-bit row offsets (frames < 2 GiB: host-checked)
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint64_t blocks_x = cdiv(cdiv(L.ow, seg_w), 4);
        uint64_t rpb = (uint64_t)L.oh * blocks_x * n / 2048; // a couple of thousand blocks per launch ...
        rpb = rpb < 16 ? 16 : (rpb > 64 ? 64 : rpb);          // ... each tall enough to amortise its window fill
        const dim3 block(kWave, 4), grid((uint32_t)blocks_x, cdiv(L.oh, (uint32_t)rpb), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
#define NUS_RD(E, C, H)                                                                                                     \
    hipLaunchKernelGGL((k_resize_down<E, C, H>), grid, block, lds, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx, T.lz_ly, \
                       T.lz_down_rows, T.lz_down_done, T.lz_stride, L.iw, L.ih, L.ow, L.oh, (uint32_t)rpb, ncols_max, ipx,  \
                       opx, L.in_sel, seg_w)
#define NUS_RD2(E, H)                        \
    switch (vc) {                            \
    case 1: NUS_RD(E, 1, H); break;          \
    case 2: NUS_RD(E, 2, H); break;          \
    case 3: NUS_RD(E, 3, H); break;          \
    case 4: NUS_RD(E, 4, H); break;          \
    default: NUS_RD(E, 5, H); break;         \
    }
        if (exact) {
            if (wide) { NUS_RD2(true, 32) } else { NUS_RD2(true, 16) }
        } else {
            if (wide) { NUS_RD2(false, 32) } else { NUS_RD2(false, 16) }
        }
#undef NUS_RD2
#undef NUS_RD
    });
}

} // namespace nus

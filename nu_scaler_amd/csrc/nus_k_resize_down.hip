// nus_k_resize_down.hip -- separable resize for DOWN-scaling factors (tap windows of 9..31 rows): the
// filters of image 0.24.9's imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251 and
// Nu_scale/src/capture/common.rs:56, :359-360 (captured frames are resized to the target with them).
#include "nus_device.hpp"
#include "nus_kernels.hpp"

namespace nus {

namespace {

constexpr uint32_t kDownSlack = 32; // zeroed LDS entries behind the row: the fixed-length horizontal loop reads up to HT - 1 past a window

// Streaming form of vertical_sample -> horizontal_sample for ratio >= 1.  On a down-scale every input
// row feeds up to 7 output rows (window 6*ratio+1 rows, outputs ratio rows apart), so instead of
// gathering a 13..31-row window per output row, a wave walks DOWN THE INPUT ROWS ONCE and keeps the
// vertical sums of the output rows in flight in registers:
//   * lane l holds VC input columns (l, l+64, ..) of the wave's 64-output segment footprint;
//   * NS = 7 accumulator slots; slot s serves output rows y_begin+s, +7, +14, ...  For each input row
//     r and slot s with r inside the slot's window:  acc[s] += wy[y][r - ly[y]] * row r  -- taps are
//     added in increasing row order, exactly the order of the per-output loop (bit-identical in
//     EXACT mode, where mac is mul + add);
//   * a row that completes a window (one per input row in the interior; the windows cut by the bottom
//     border end together on the last row) writes its sums to the wave's LDS row; each lane then sums
//     its output's horizontal taps from LDS, packs and stores; the slot restarts on its next output row.
// Every input byte is read once per row block (+ the window fill of the block's first rows), every
// vertical product is computed once, and no slot ever holds more than one open window (host-checked:
// ly[y+7] > ly[y] + ny[y] - 1).
constexpr int kDownSlots = 7;

template <bool EXACT, int VC, int HT>
__global__ __launch_bounds__(256) void k_resize_down(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const int32_t *__restrict__ lxt, const uint32_t *__restrict__ nxt, const float *__restrict__ wxt,
    const int32_t *__restrict__ lyt, const uint32_t *__restrict__ nyt, const float *__restrict__ wyt,
    uint32_t stride, uint32_t iw, uint32_t ow, uint32_t oh, uint32_t rows_per_block, uint32_t ncols_max,
    size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = kDownSlots;
    float4 *s_v = reinterpret_cast<float4 *>(smem) + (size_t)threadIdx.y * (ncols_max + kDownSlack);
    const uint32_t seg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + threadIdx.y);
    const uint32_t X0 = seg * kWave;
    if (X0 >= ow) return; // whole wave; no workgroup barriers below
    const uint32_t Xlast = umin(X0 + kWave, ow) - 1;
    const int32_t cmin = lxt[X0];
    const int32_t ncols = lxt[Xlast] + (int32_t)nxt[Xlast] - cmin; // <= 64 * VC (host-checked)
    if (threadIdx.x < kDownSlack) s_v[ncols + threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t x = X0 + threadIdx.x;
    const bool lane_active = x < ow;
    const uint32_t xo = lane_active ? x : X0;
    const int32_t hl = lxt[xo] - cmin;
    // the lane's horizontal weights, zero beyond its window (HT >= the widest window, host-checked); an adding of
    // +-0 leaves every sum as it is, so the fixed trip count does not change a bit
    float hw[HT];
    {
        const uint32_t hn = nxt[xo];
        const float *wx = wxt + (size_t)xo * stride;
#pragma unroll
        for (int k = 0; k < HT; ++k) hw[k] = (uint32_t)k < hn ? wx[(uint32_t)k < stride ? k : 0] : 0.0f;
    }
    const uint32_t y_begin = blockIdx.y * rows_per_block;
    const uint32_t y_end = umin(y_begin + rows_per_block, oh);
    const uint32_t *base = in + (size_t)blockIdx.z * in_frame_px;
    uint32_t *dst = out + (size_t)blockIdx.z * out_frame_px + x;

    // this lane's VC input columns: lane, lane + 64, ...: each load of the wave is one contiguous 256 B and
    // its LDS writes are 64 consecutive float4; clamped into the row (columns past the footprint are never read back)
    uint32_t col[VC];
#pragma unroll
    for (int m = 0; m < VC; ++m) col[m] = umin((uint32_t)cmin + threadIdx.x + kWave * m, iw - 1);
    auto load_row = [&](int32_t r, uint32_t (&raw)[VC]) {
        const uint32_t *row = base + (size_t)r * iw;
#pragma unroll
        for (int m = 0; m < VC; ++m) raw[m] = row[col[m]];
    };

    // slots: output row, first tap row, tap count, offset of the row's weights (wave-uniform -> scalar registers)
    uint32_t sy[NS], sn[NS], so[NS];
    int32_t sl[NS];
    float acc[NS][VC * 4];
    auto open_slot = [&](int s, uint32_t y) {
        sy[s] = y;
        const uint32_t yc = umin(y, oh - 1);
        sl[s] = lyt[yc];
        sn[s] = y < y_end ? nyt[yc] : 0u; // past the block: never inside
        so[s] = yc * stride;
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        open_slot(s, y_begin + s);
#pragma unroll
        for (int k = 0; k < VC * 4; ++k) acc[s][k] = 0.0f;
    }
    const int32_t r_first = sl[0];
    const int32_t r_last = lyt[y_end - 1] + (int32_t)nyt[y_end - 1] - 1;
    uint32_t raw_next[VC];
    load_row(r_first, raw_next);

    for (int32_t r = r_first; r <= r_last; ++r) {
        float p[VC * 4];
#pragma unroll
        for (int m = 0; m < VC; ++m) {
            const uint32_t px = swz(raw_next[m], sel);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[m * 4 + c] = ch_f32(px, c);
        }
        if (r < r_last) load_row(r + 1, raw_next); // one row ahead
        // this row's weight in every slot (0 where the row is outside the slot's window): scalar work and NS
        // independent scalar loads up front, then NS x VC x 4 FMAs with no branch in between
        float w[NS];
        bool last[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t k = (uint32_t)(r - sl[s]);
            const bool inside = k < sn[s]; // wave-uniform (k wraps to a huge value above the window)
            const float ws = wyt[so[s] + (inside ? k : 0u)];
            w[s] = inside ? ws : 0.0f;
            last[s] = inside && k + 1 == sn[s]; // (a closed slot has sn == 0 and is never inside)
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            asm volatile("" : "+v"(w[s])); // VGPR copy: scalar operands halve the VALU issue rate
#pragma unroll
            for (int q = 0; q < VC * 4; ++q) acc[s][q] = mac<EXACT>(acc[s][q], p[q], w[s]);
        }
        // completed windows (one per input row in the interior; the rows cut by the bottom border end together):
        // sums -> LDS row, horizontal pass, store; the slot restarts NS output rows further down
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!last[s]) continue; // wave-uniform
#pragma unroll
            for (int m = 0; m < VC; ++m) {
                const int32_t ci = (int32_t)threadIdx.x + kWave * m;
                if (ci < ncols) s_v[ci] = make_float4(acc[s][m * 4], acc[s][m * 4 + 1], acc[s][m * 4 + 2], acc[s][m * 4 + 3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[s][m * 4 + c] = 0.0f;
            }
            const uint32_t y = sy[s];
            open_slot(s, y + NS);
            // A wave only ever reads the LDS row it wrote itself, and the LDS executes one wave's
            // instructions in order: no workgroup barrier, just keep the compiler from reordering.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
#pragma unroll
            for (int k = 0; k < HT; ++k) { // taps beyond the lane's window: weight 0 times a finite LDS value
                const float4 v = s_v[hl + k];
                h0 = mac<EXACT>(h0, v.x, hw[k]);
                h1 = mac<EXACT>(h1, v.y, hw[k]);
                h2 = mac<EXACT>(h2, v.z, hw[k]);
                h3 = mac<EXACT>(h3, v.w, hw[k]);
            }
            if (lane_active)
                dst[(size_t)y * ow] = pack_u8<EXACT>(h3, 3, pack_u8<EXACT>(h2, 2, pack_u8<EXACT>(h1, 1, pack_u8<EXACT>(h0, 0, 0u))));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next completed row overwrites s_v
            __builtin_amdgcn_wave_barrier();
        }
    }
}

} // namespace

hipError_t launch_resize_down(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max, uint32_t max_taps_x)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    const uint32_t vc = cdiv(ncols_max, kWave);
    if (vc < 1 || vc > 5 || max_taps_x > 32) return hipErrorInvalidValue;
    const bool wide = max_taps_x > 16; // horizontal weights per lane: 16 or 32 registers
    const size_t lds = (size_t)4 * (ncols_max + kDownSlack) * sizeof(float4);
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint64_t blocks_x = cdiv(cdiv(L.ow, kWave), 4);
        uint64_t rpb = (uint64_t)L.oh * blocks_x * n / 2048; // a couple of thousand blocks per launch ...
        rpb = rpb < 16 ? 16 : (rpb > 64 ? 64 : rpb);          // ... each tall enough to amortise its window fill
        const dim3 block(kWave, 4), grid((uint32_t)blocks_x, cdiv(L.oh, (uint32_t)rpb), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
#define NUS_RD(E, C, H)                                                                                                     \
    hipLaunchKernelGGL((k_resize_down<E, C, H>), grid, block, lds, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx, T.lz_ly, \
                       T.lz_ny, T.lz_wy, T.lz_stride, L.iw, L.ow, L.oh, (uint32_t)rpb, ncols_max, ipx, opx, L.in_sel)
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

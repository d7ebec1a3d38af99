// nus_k_resize.hip -- the any-scale separable resize kernels for up-scaling and mixed factors (gfx950, wave64):
// per-pixel fallback, LDS-row kernel, register-window kernel.  (Down-scaling: nus_k_resize_down.hip; exact x2 / x4:
// nus_k_lanczos_x2.hip / nus_k_lanczos_xs.hip.)
//
// Reference arithmetic being reproduced (paths relative to the reference checkout):
//   lanczos3 / catmull-rom / triangle
//              image-0.24.9 imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251
#include "nus_device.hpp"

namespace nus {

namespace {

// ---------------------------------------------------------------------------------
// Lanczos-3 (image-0.24.9 resize: vertical pass into f32, then horizontal pass)
// ---------------------------------------------------------------------------------

// Any scale, one output pixel per thread: for every horizontal tap column the vertical
// sum is formed first (f32, tap order ascending), then the horizontal sum, exactly the
// operation order of the two-pass CPU algorithm.  Also used for the first/last
// `edge_cols` output columns next to the x2 kernel, whose interior weights do not
// apply there.  blockDim = (64, 4).
template <bool EXACT>
__global__ __launch_bounds__(256) void k_lanczos_general(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const int32_t *__restrict__ lxt, const uint32_t *__restrict__ nxt, const float *__restrict__ wxt,
    const int32_t *__restrict__ lyt, const uint32_t *__restrict__ nyt, const float *__restrict__ wyt,
    uint32_t stride, uint32_t iw, uint32_t ow, uint32_t oh, uint32_t ncols, uint32_t split, uint32_t gap,
    size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    const uint32_t y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (y >= oh || i >= ncols) return;
    const uint32_t x = i < split ? i : i + gap;
    const int32_t lx = lxt[x];
    const uint32_t nx = nxt[x];
    const int32_t ly = lyt[y];
    const uint32_t ny = nyt[y];
    const float *wx = wxt + (size_t)x * stride;
    const float *wy = wyt + (size_t)y * stride;
    const uint32_t *src = in + (size_t)blockIdx.z * in_frame_px + (size_t)ly * iw + lx;
    float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
    for (uint32_t a = 0; a < nx; ++a) {
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
        for (uint32_t b = 0; b < ny; ++b) {
            const uint32_t p = swz(src[(size_t)b * iw + a], sel);
            const float w = wy[b];
            v0 = mac<EXACT>(v0, ch_f32(p, 0), w);
            v1 = mac<EXACT>(v1, ch_f32(p, 1), w);
            v2 = mac<EXACT>(v2, ch_f32(p, 2), w);
            v3 = mac<EXACT>(v3, ch_f32(p, 3), w);
        }
        const float w = wx[a];
        h0 = mac<EXACT>(h0, v0, w);
        h1 = mac<EXACT>(h1, v1, w);
        h2 = mac<EXACT>(h2, v2, w);
        h3 = mac<EXACT>(h3, v3, w);
    }
    out[(size_t)blockIdx.z * out_frame_px + (size_t)y * ow + x] =
        pack_u8<EXACT>(h3, 3, pack_u8<EXACT>(h2, 2, pack_u8<EXACT>(h1, 1, pack_u8<EXACT>(h0, 0, 0u))));
}

// Any scale, separable, two passes per output row through an LDS row (the data flow of
// vertical_sample -> horizontal_sample with only ONE f32 row of the intermediate image alive):
//   blockDim = (64, 4): the 4 waves own 4 adjacent output column segments (64*N columns each) of the
//   same block of output rows; each wave has its own LDS row and never reads another wave's.
//   per output row y:  V pass -- the lanes sweep the input columns their segment's taps touch and
//                      store  V[col] = sum_j wy[y][j] * in[ly[y]+j][col]  (f32 x 4 channels) in LDS;
//                      H pass -- each lane sums its outputs' taps from LDS (16-B reads), packs, stores.
// Same f32 operation order as k_lanczos_general (and the CPU algorithm); replaces its nx*ny taps per
// pixel by nx + ny/scale.  SMALL: every window has <= 8 taps (any upscale), weights stay in VGPRs.
// UNION > 0 (needs VEC and SMALL): the tap windows of a lane's 4 adjacent outputs overlap almost
// completely on an upscale, so the H pass reads their UNION (<= UNION columns) from LDS once into
// registers and gives every output a UNION-long weight vector that is zero outside its own window --
// 10-12 LDS reads per lane per row instead of 32 (the LDS pipe was the limiter), for ~40 % more FMAs
// whose extra terms are exact +-0.  Same order of the non-zero terms, so the bits do not change.
constexpr uint32_t kResizeSlack = 16; // zeroed LDS entries behind each row: windows may read past its end

template <bool EXACT, bool VEC, bool SMALL, int UNION>
__global__ __launch_bounds__(256) void k_resize_rows(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const int32_t *__restrict__ lxt, const uint32_t *__restrict__ nxt, const float *__restrict__ wxt,
    const int32_t *__restrict__ lyt, const uint32_t *__restrict__ nyt, const float *__restrict__ wyt,
    uint32_t stride, uint32_t iw, uint32_t ow, uint32_t oh, uint32_t rows_per_block, uint32_t ncols_max,
    size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int N = VEC ? 4 : 1;
    constexpr uint32_t SEGW = kWave * N;
    static_assert(UNION == 0 || (VEC && SMALL), "the union-window H pass needs 4 outputs per lane and <= 8 taps");
    float4 *s_v = reinterpret_cast<float4 *>(smem) + (size_t)threadIdx.y * (ncols_max + kResizeSlack);
    const uint32_t seg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + threadIdx.y);
    const uint32_t X0 = seg * SEGW;
    if (X0 >= ow) return; // whole wave; no workgroup barriers below
    const uint32_t Xlast = umin(X0 + SEGW, ow) - 1;
    const int32_t cmin = lxt[X0];
    const int32_t cmax = lxt[Xlast] + (int32_t)nxt[Xlast];
    const uint32_t x = X0 + threadIdx.x * N;
    const bool lane_active = x < ow;
    const uint32_t y_begin = blockIdx.y * rows_per_block;
    const uint32_t y_end = umin(y_begin + rows_per_block, oh);
    const uint32_t *base = in + (size_t)blockIdx.z * in_frame_px;
    uint32_t *dst = out + (size_t)blockIdx.z * out_frame_px + x;

    if (threadIdx.x < kResizeSlack) s_v[(cmax - cmin) + threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // slack, see the H pass
    // horizontal windows of this lane's outputs
    int32_t hl[N];
    uint32_t hn[N];
    float hw[N][8];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t xo = lane_active ? x + i : 0;
        hl[i] = lxt[xo] - cmin;
        hn[i] = nxt[xo];
        if (SMALL && UNION == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) hw[i][k] = wxt[(size_t)xo * stride + k]; // zero padded beyond hn
        }
    }
    // union window: weights of output i re-based to the first output's left column
    constexpr int UW = UNION > 0 ? UNION : 1;
    float hu[N][UW];
    if (UNION > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t xo = lane_active ? x + i : 0;
            const int32_t shift = hl[i] - hl[0]; // >= 0: left edges do not decrease with x
#pragma unroll
            for (int j = 0; j < UW; ++j) {
                const int32_t k = j - shift;
                const float w = wxt[(size_t)xo * stride + (uint32_t)(k < 0 ? 0 : (k > 7 ? 7 : k))];
                hu[i][j] = (k >= 0 && k < 8) ? w : 0.0f;
            }
        }
    }

    for (uint32_t y = y_begin; y < y_end; ++y) {
        const int32_t ly = lyt[y];
        const uint32_t ny = nyt[y];
        const float *wy = wyt + (size_t)y * stride;
        {
            float wv[8];
            if (SMALL) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    wv[j] = wy[j];
                    asm volatile("" : "+v"(wv[j])); // VGPR copy: scalar operands halve the VALU issue rate
                }
            }
            const uint32_t *src = base + (size_t)ly * iw;
            // V pass: 4 input columns per lane per sweep; all tap rows of a group are requested before
            // the first is consumed (16-B loads where the group lies inside the row)
            for (int32_t col = cmin + 4 * (int32_t)threadIdx.x; col < cmax; col += 4 * kWave) {
                float v[4][4] = {{0.0f}};
                const bool whole = col + 4 <= (int32_t)iw; // else: last group of the row, per-pixel loads
                if (SMALL) {
                    uint32_t p[8][4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if ((uint32_t)j < ny) { // wave-uniform
                            const uint32_t *q = src + (size_t)j * iw + col;
                            if (whole) {
                                // dword-aligned 16-B load (global loads need no 16-B alignment)
                                const uint4 t = *reinterpret_cast<const __attribute__((aligned(4))) uint4 *>(q);
                                p[j][0] = t.x; p[j][1] = t.y; p[j][2] = t.z; p[j][3] = t.w;
                            } else {
#pragma unroll
                                for (int m = 0; m < 4; ++m) p[j][m] = col + m < (int32_t)iw ? q[m] : 0u;
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if ((uint32_t)j < ny) {
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int c = 0; c < 4; ++c) v[m][c] = mac<EXACT>(v[m][c], ch_f32(swz(p[j][m], sel), c), wv[j]);
                        }
                    }
                } else {
                    for (uint32_t j = 0; j < ny; ++j) {
                        const uint32_t *q = src + (size_t)j * iw + col;
                        const float w = wy[j];
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const uint32_t px = swz(col + m < (int32_t)iw ? q[m] : 0u, sel);
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[m][c] = mac<EXACT>(v[m][c], ch_f32(px, c), w);
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (col + m < cmax) s_v[col - cmin + m] = make_float4(v[m][0], v[m][1], v[m][2], v[m][3]);
            }
        }
        // A wave only ever reads the LDS row it wrote itself, and the LDS executes one wave's
        // instructions in order: no workgroup barrier, just keep the compiler from reordering.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane_active) {
            uint32_t o[N];
            float4 R[UW];
            if (UNION > 0) {
#pragma unroll
                for (int j = 0; j < UW; ++j) R[j] = s_v[hl[0] + j];
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
                if (UNION > 0) {
#pragma unroll
                    for (int j = 0; j < UW; ++j) {
                        h0 = mac<EXACT>(h0, R[j].x, hu[i][j]);
                        h1 = mac<EXACT>(h1, R[j].y, hu[i][j]);
                        h2 = mac<EXACT>(h2, R[j].z, hu[i][j]);
                        h3 = mac<EXACT>(h3, R[j].w, hu[i][j]);
                    }
                } else if (SMALL) {
                    // all 8 slots, no per-lane branch: slots beyond the window carry weight 0 and read
                    // finite values (the row has 8 zeroed slack entries), so they add +-0
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 v = s_v[hl[i] + k];
                        h0 = mac<EXACT>(h0, v.x, hw[i][k]);
                        h1 = mac<EXACT>(h1, v.y, hw[i][k]);
                        h2 = mac<EXACT>(h2, v.z, hw[i][k]);
                        h3 = mac<EXACT>(h3, v.w, hw[i][k]);
                    }
                } else {
                    const float *wx = wxt + (size_t)(x + i) * stride;
                    for (uint32_t k = 0; k < hn[i]; ++k) {
                        const float4 v = s_v[hl[i] + (int32_t)k];
                        const float w = wx[k];
                        h0 = mac<EXACT>(h0, v.x, w);
                        h1 = mac<EXACT>(h1, v.y, w);
                        h2 = mac<EXACT>(h2, v.z, w);
                        h3 = mac<EXACT>(h3, v.w, w);
                    }
                }
                o[i] = pack_u8<EXACT>(h3, 3, pack_u8<EXACT>(h2, 2, pack_u8<EXACT>(h1, 1, pack_u8<EXACT>(h0, 0, 0u))));
            }
            if (VEC)
                *reinterpret_cast<uint4 *>(dst + (size_t)y * ow) = make_uint4(o[0], o[N > 1 ? 1 : 0], o[N > 2 ? 2 : 0], o[N > 3 ? 3 : 0]);
            else
                dst[(size_t)y * ow] = o[0];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next row's V pass overwrites s_v
        __builtin_amdgcn_wave_barrier();
    }
}

// Up-scaling variant of k_resize_rows with the vertical taps served from registers, as in the x2 kernel:
// on an upscale the first tap row ly[y] advances by 0 or 1 per output row (host-checked), so each lane
// keeps a WR-row f32 window of its VC input columns whose row 0 is always ly[y]; an advance shifts the
// window by one row (register moves) and converts the one new row, prefetched a step ahead.  The V
// pass is then WR FMAs per value with no load, no u8->f32 convert and no index arithmetic in the chain;
// rows of the window beyond the tap count carry weight 0 (the table is zero padded) and hold finite
// pixels (row index clamped), so they add +-0.  H pass and LDS row exactly as in k_resize_rows.
// Needs: ow % 4 == 0, <= WR vertical and <= 8 horizontal taps, segment footprint <= 64 * VC columns.
// N outputs per lane: 4 (segments of 256 output columns; factors >= ~x1.4) or 2 (segments of 128: the
// footprint of factors x1.0 .. x1.4 then still fits 3 columns per lane).
constexpr int kResizeWinRows = 7; // Lanczos-3 on an upscale touches at most 7 input rows

// Row prefetch of k_resize_win (round 3): as in k_resize_down the rows come through a per-wave LDS-DMA ring -- VC pieces of 64
// pixels per row, requested kWinDepth window advances ahead from inline assembly (no VGPR destination, invisible to the
// compiler's s_waitcnt insertion) -- and are waited for with a hand-placed `s_waitcnt vmcnt(N)`.  With ordinary loads the compiler
// waited for vmcnt(0) -- every store of the wave -- once per output row (tools/check_hidden_loads.py lists such waits).
#ifndef NUS_WIN_DEPTH
#define NUS_WIN_DEPTH 2
#endif
constexpr int kWinDepth = NUS_WIN_DEPTH;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in this kernel uses it
__device__ __forceinline__ void win_dma_row4(const void *base, uint32_t off, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N, int BACK>
__device__ __forceinline__ void win_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}

template <bool EXACT, int VC, int UNION, int N>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((UNION > 0 && VC == 3) ? 3 : 1))) void k_resize_win(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const int32_t *__restrict__ lxt, const uint32_t *__restrict__ nxt, const float *__restrict__ wxt,
    const int32_t *__restrict__ lyt, const float *__restrict__ wyt,
    uint32_t stride, uint32_t iw, uint32_t ih, uint32_t ow, uint32_t oh, uint32_t rows_per_block, uint32_t ncols_max,
    size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WR = kResizeWinRows;
    static_assert(WR == 7, "the walk below is written out for seven window positions");
    static_assert(N == 4 || N == 2, "outputs per lane");
    constexpr uint32_t SEGW = kWave * N;
    // The ring serves every shape but the one whose union weights live in LDS (3 columns per lane + union H pass: 51 KB per block,
    // three blocks per CU; 6 KB of ring would leave room for two: 1080p -> 1800p 18.4 -> 21.9 us).  That shape keeps the next row
    // in registers, requested one advance ahead with ordinary loads.
    constexpr bool RING = !(UNION > 0 && VC == 3);
    constexpr int D = RING ? kWinDepth : 0;
    // the four waves' row rings first (LDS-DMA takes its LDS address from M0: keep it a small offset), then the rest
    constexpr size_t ring_bytes = (size_t)4 * D * VC * kWave * sizeof(uint32_t);
    uint32_t *s_ring = reinterpret_cast<uint32_t *>(smem) + (size_t)threadIdx.y * (D * VC * kWave);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_ring);
    float4 *s_v = reinterpret_cast<float4 *>(smem + ring_bytes) + (size_t)threadIdx.y * (ncols_max + kResizeSlack);
    const GridPos g = xcd_contiguous_pos(); // row blocks that share their window-fill rows behind one L2
    const uint32_t seg = __builtin_amdgcn_readfirstlane(g.x * 4 + threadIdx.y);
    const uint32_t X0 = seg * SEGW;
    if (X0 >= ow) return; // whole wave; no workgroup barriers below
    const uint32_t Xlast = umin(X0 + SEGW, ow) - 1;
    const int32_t cmin = lxt[X0];
    const int32_t ncols = lxt[Xlast] + (int32_t)nxt[Xlast] - cmin; // <= 64 * VC (host-checked)
    const uint32_t x = X0 + threadIdx.x * N;
    const bool lane_active = x < ow;
    const uint32_t y_begin = g.y * rows_per_block;
    const uint32_t y_end = umin(y_begin + rows_per_block, oh);
    const uint32_t *base = in + (size_t)g.z * in_frame_px;
    uint32_t *dst = out + (size_t)g.z * out_frame_px + x;

    if (threadIdx.x < kResizeSlack) s_v[ncols + threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // horizontal windows of this lane's 4 outputs
    int32_t hl[N];
    float hw[UNION > 0 ? 1 : N][8];
    constexpr int UW = UNION > 0 ? UNION : 1;
    // With 3 columns per lane the window (84 VGPRs), the union window (40) and the N x UW union weights do not fit three
    // waves per SIMD: the weights then live in LDS ([tap][lane] float4 over the lane's outputs, behind the four waves'
    // rows) and the H pass walks the taps, one 16-byte read of the V row and one of the weights per tap, all N outputs
    // accumulating side by side (each output still receives its terms in tap order: same bits).
    constexpr bool HU_LDS = UNION > 0 && VC == 3;
    float hu[(UNION > 0 && !HU_LDS) ? N : 1][UW];
    float4 *const s_hu = reinterpret_cast<float4 *>(smem + ring_bytes) + (size_t)4 * (ncols_max + kResizeSlack) +
                         (size_t)threadIdx.y * (UW * kWave) + threadIdx.x;
    float wtmp[N][UW];
#pragma unroll
    for (int i = 0; i < N; ++i) hl[i] = lxt[lane_active ? x + i : 0] - cmin;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t xo = lane_active ? x + i : 0;
        if (UNION > 0) {
            const int32_t shift = hl[i] - hl[0];
#pragma unroll
            for (int j = 0; j < UW; ++j) {
                const int32_t k = j - shift;
                const float w = wxt[(size_t)xo * stride + (uint32_t)(k < 0 ? 0 : (k > 7 ? 7 : k))];
                const float wj = (k >= 0 && k < 8) ? w : 0.0f;
                if (HU_LDS) wtmp[i][j] = wj;
                else hu[HU_LDS ? 0 : i][j] = wj;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) hw[i][k] = wxt[(size_t)xo * stride + k];
        }
    }

    if (HU_LDS) {
#pragma unroll
        for (int j = 0; j < UW; ++j)
            s_hu[j * kWave] = make_float4(wtmp[0][j], wtmp[1][j], N > 2 ? wtmp[N > 2 ? 2 : 0][j] : 0.0f, N > 3 ? wtmp[N > 3 ? 3 : 0][j] : 0.0f);
    }

    // this lane's VC input columns: lane, lane + 64, ... of the footprint, so that for each m the wave's loads
    // are one contiguous 256 B and its LDS writes 64 consecutive float4 (no bank conflict); clamped into the
    // row -- columns past the footprint are never read back
    uint32_t col[VC];
#pragma unroll
    for (int m = 0; m < VC; ++m) col[m] = umin((uint32_t)cmin + threadIdx.x + kWave * m, iw - 1);
    auto load_row = [&](int32_t r, uint32_t (&raw)[VC]) {
        const uint32_t rr = (uint32_t)(r < 0 ? 0 : (r > (int32_t)ih - 1 ? (int32_t)ih - 1 : r));
        const uint32_t *row = base + (size_t)rr * iw;
#pragma unroll
        for (int m = 0; m < VC; ++m) raw[m] = row[col[m]];
    };
    auto cvt = [&](const uint32_t (&raw)[VC], float (&dstrow)[VC * 4]) {
#pragma unroll
        for (int m = 0; m < VC; ++m) {
            const uint32_t p = swz(raw[m], sel);
#pragma unroll
            for (int c = 0; c < 4; ++c) dstrow[m * 4 + c] = ch_f32(p, c);
        }
    };

    // FMA mode: rows whose pixels are all opaque in this wave (bit j of `opq` = window row WR-1-j); an output
    // row whose whole window is opaque skips the alpha channel and stores 255, as in the x2 kernel
    constexpr bool OP = !EXACT;
    auto row_opaque = [&](const uint32_t (&raw)[VC]) -> uint32_t {
        uint32_t a = 0xFFFFFFFFu;
#pragma unroll
        for (int m = 0; m < VC; ++m) a &= swz(raw[m], sel);
        return __builtin_amdgcn_ballot_w64(a < 0xFF000000u) == 0ull ? 1u : 0u;
    };
    // (scalar loads of the per-row table: a vector load + readfirstlane makes the compiler wait for vmcnt(0) every row)
    typedef const __attribute__((address_space(4))) int32_t *ci32_p;
    ci32_p lyt_s = (ci32_p)(uintptr_t)lyt;
    int32_t top = lyt_s[y_begin];
    uint32_t col_off[VC]; // byte offset of the lane's column m inside a row
#pragma unroll
    for (int m = 0; m < VC; ++m) col_off[m] = col[m] * 4u;
    auto request_row = [&](int32_t r, uint32_t slot) {
        const uint32_t rr = (uint32_t)(r < 0 ? 0 : (r > (int32_t)ih - 1 ? (int32_t)ih - 1 : r));
#pragma unroll
        for (int m = 0; m < VC; ++m) win_dma_row4(base, rr * (iw * 4u) + col_off[m], ring_lds + (slot * VC + m) * (kWave * 4u));
    };
    float win[WR][VC * 4];
    uint32_t opq = 0;
    {
        uint32_t raw[WR][VC];
#pragma unroll
        for (int j = 0; j < WR; ++j) load_row(top + j, raw[j]);
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            if (OP) opq = (opq << 1) | row_opaque(raw[j]);
            cvt(raw[j], win[j]);
        }
    }
    // rows top + WR .. top + WR + D - 1, requested D advances ahead; the hand-placed waits of the loop assume that nothing older
    // than these requests is outstanding
    uint32_t next[VC]; // !RING: row top + WR, requested one advance ahead
    if (RING) {
#pragma unroll
        for (int k = 0; k < D; ++k) request_row(top + WR + k, (uint32_t)k);
        win_wait_vmcnt<0, 0>(); // (the count below needs the D stores that follow a request: not yet there for these first rows)
    } else {
        load_row(top + WR, next);
    }
    uint32_t adv = 0; // window advances so far: row top + WR sits in ring slot adv % D

    // The window ROTATES (round 5; rounds 2 - 4 shifted it down by one row per advance: 6 x VC x 4 register moves in front of the row's
    // arithmetic, a quarter of the kernel's time at x1.1 - x1.4, profiles/r05_resize_win_rotating_window.txt): row top + j sits in slot
    // (R + j) % WR, and the row that leaves gives its slot to the row that enters.  R must be a compile-time constant (the window is
    // registers), so the walk is written as one turn of the window -- WR positions, each with its output rows (at least one: the first
    // tap row advances by 0 or 1 per output row, host-checked) and the advance behind them -- repeated until the block's rows are done.
    uint32_t y = y_begin;
    int32_t ly_next = top;
    auto advance = [&](auto rc, const int32_t ly) __attribute__((always_inline)) { // wave-uniform; ly == top + 1 (host-checked)
        constexpr int R = decltype(rc)::value;
        // Issued since the request of row top + WR: the VC pieces of the D - 1 rows behind it and at least one store per
        // output row -- and an advance happens at most once per output row, so at least D stores: with (D - 1) VC + D
        // instructions allowed outstanding the row has landed (more stores in between make the wait stricter, never looser)
        if (RING) {
            const uint32_t slot = adv % (uint32_t)(D > 0 ? D : 1);
            win_wait_vmcnt<(D > 0 ? (D - 1) * VC + D : 0), (D > 0 ? (D - 1) * VC + 1 : 1)>();
#pragma unroll
            for (int m = 0; m < VC; ++m) next[m] = s_ring[(slot * VC + m) * kWave + threadIdx.x];
            // the slot is requested again only when its reads have RETURNED (nothing orders a queued ds_read behind a later
            // LDS-DMA write: see k_resize_down)
            asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
            request_row(ly + WR + D - 1, slot);
            ++adv;
        }
        if (OP) opq = (opq << 1) | row_opaque(next);
        cvt(next, win[R]); // the slot of the row that leaves
        top = ly;
        if (!RING) load_row(top + WR, next);
    };
    auto output_row = [&](auto rc) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
        // the NEXT output row's first tap row, asked for now: the position's loop condition needs it when this row is done
        ly_next = lyt_s[y + 1 < y_end ? y + 1 : y];
        // (scalar loads, spelled out: with the row counter shared between the window positions' loops the compiler no longer sees
        // that the address is wave-uniform, and vector loads here cost a drain of the wave's stores per row)
        const size_t wy = (size_t)__builtin_amdgcn_readfirstlane(y) * stride;
        float wv[WR];
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            wv[j] = uniform_load(wyt, wy + j); // zero padded beyond the row's tap count
            asm volatile("" : "+v"(wv[j])); // VGPR copy: scalar operands halve the VALU issue rate
        }
        const bool skip_alpha = OP && (opq & ((1u << WR) - 1u)) == ((1u << WR) - 1u); // wave-uniform
#pragma unroll
        for (int m = 0; m < VC; ++m) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c == 3 && skip_alpha) {
                    v[3] = 0.0f;
                    continue;
                }
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < WR; ++j) acc = mac<EXACT>(acc, win[(R + j) % WR][m * 4 + c], wv[j]);
                v[c] = acc;
            }
            const int32_t ci = (int32_t)threadIdx.x + kWave * m;
            if (ci < ncols) s_v[ci] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane_active) {
            uint32_t o[N];
            if (UNION > 0 && (HU_LDS || EXACT)) { // (FMA mode with register weights: the output-outer form below allocates better)
                // tap-outer: one 16-byte read of the V row per tap, all N outputs accumulating side by side (each output
                // still receives its terms in tap order: same bits); nothing but the accumulators stays live
                float h[N][4] = {{0.0f}};
#pragma unroll
                for (int j = 0; j < UW; ++j) {
                    const float4 r = s_v[hl[0] + j];
                    float wi[4];
                    if (HU_LDS) {
                        const float4 w4 = s_hu[j * kWave];
                        wi[0] = w4.x, wi[1] = w4.y, wi[2] = w4.z, wi[3] = w4.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < N; ++i) wi[i] = hu[HU_LDS ? 0 : i][j];
                    }
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        h[i][0] = mac<EXACT>(h[i][0], r.x, wi[i]);
                        h[i][1] = mac<EXACT>(h[i][1], r.y, wi[i]);
                        h[i][2] = mac<EXACT>(h[i][2], r.z, wi[i]);
                        if (!skip_alpha) h[i][3] = mac<EXACT>(h[i][3], r.w, wi[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const uint32_t rgb = pack_u8<EXACT>(h[i][2], 2, pack_u8<EXACT>(h[i][1], 1, pack_u8<EXACT>(h[i][0], 0, 0u)));
                    o[i] = skip_alpha ? (rgb | 0xFF000000u) : pack_u8<EXACT>(h[i][3], 3, rgb);
                }
            } else {
            float4 R[UW];
            if (UNION > 0) {
#pragma unroll
                for (int j = 0; j < UW; ++j) R[j] = s_v[hl[0] + j];
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
                if (UNION > 0) {
#pragma unroll
                    for (int j = 0; j < UW; ++j) {
                        h0 = mac<EXACT>(h0, R[j].x, hu[HU_LDS ? 0 : i][j]);
                        h1 = mac<EXACT>(h1, R[j].y, hu[HU_LDS ? 0 : i][j]);
                        h2 = mac<EXACT>(h2, R[j].z, hu[HU_LDS ? 0 : i][j]);
                    }
                    if (!skip_alpha) {
#pragma unroll
                        for (int j = 0; j < UW; ++j) h3 = mac<EXACT>(h3, R[j].w, hu[HU_LDS ? 0 : i][j]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 v = s_v[hl[i] + k];
                        h0 = mac<EXACT>(h0, v.x, hw[i][k]);
                        h1 = mac<EXACT>(h1, v.y, hw[i][k]);
                        h2 = mac<EXACT>(h2, v.z, hw[i][k]);
                        if (!skip_alpha) h3 = mac<EXACT>(h3, v.w, hw[i][k]);
                    }
                }
                const uint32_t rgb = pack_u8<EXACT>(h2, 2, pack_u8<EXACT>(h1, 1, pack_u8<EXACT>(h0, 0, 0u)));
                o[i] = skip_alpha ? (rgb | 0xFF000000u) : pack_u8<EXACT>(h3, 3, rgb);
            }
            }
            if constexpr (N == 4)
                *reinterpret_cast<uint4 *>(dst + (size_t)y * ow) = make_uint4(o[0], o[1], o[2], o[3]);
            else
                *reinterpret_cast<uint2 *>(dst + (size_t)y * ow) = make_uint2(o[0], o[1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next row's V pass overwrites s_v
        __builtin_amdgcn_wave_barrier();
    };
#define NUS_WIN_POSITION(R)                                          \
    do {                                                             \
        output_row(std::integral_constant<int, R>{});                \
        ++y;                                                         \
    } while (y < y_end && ly_next == top);                           \
    if (y >= y_end) break;                                           \
    advance(std::integral_constant<int, R>{}, ly_next);
    for (;;) {
        NUS_WIN_POSITION(0)
        NUS_WIN_POSITION(1)
        NUS_WIN_POSITION(2)
        NUS_WIN_POSITION(3)
        NUS_WIN_POSITION(4)
        NUS_WIN_POSITION(5)
        NUS_WIN_POSITION(6)
    }
#undef NUS_WIN_POSITION
}

} // namespace

hipError_t launch_lanczos_general(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t edge_cols)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    uint32_t ncols = L.ow, split = L.ow, gap = 0;
    if (edge_cols && 2 * edge_cols < L.ow) {
        ncols = 2 * edge_cols;
        split = edge_cols;
        gap = L.ow - 2 * edge_cols;
    }
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const dim3 block(kWave, 4), grid(cdiv(ncols, 64), cdiv(L.oh, 4), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
        if (exact)
            hipLaunchKernelGGL(k_lanczos_general<true>, grid, block, 0, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx,
                               T.lz_ly, T.lz_ny, T.lz_wy, T.lz_stride, L.iw, L.ow, L.oh, ncols, split, gap, ipx, opx, L.in_sel);
        else
            hipLaunchKernelGGL(k_lanczos_general<false>, grid, block, 0, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx,
                               T.lz_ly, T.lz_ny, T.lz_wy, T.lz_stride, L.iw, L.ow, L.oh, ncols, split, gap, ipx, opx, L.in_sel);
    });
}

hipError_t launch_resize_rows(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max, bool small_taps,
                              uint32_t union_taps)
{
    const bool vec = (L.ow % 4) == 0;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    const uint32_t segw = vec ? 256 : 64;
    const size_t lds = (size_t)4 * (ncols_max + kResizeSlack) * sizeof(float4);
    // union-window H pass: widest union of a lane's 4 windows, rounded up to an instantiated size
    const int uni = (vec && small_taps && union_taps > 0 && union_taps <= 12) ? (union_taps <= 10 ? 10 : 12) : 0;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint64_t blocks_x = cdiv(cdiv(L.ow, segw), 4);
        uint64_t rpb = (uint64_t)L.oh * blocks_x * n / 4096; // a few thousand blocks per launch
        rpb = rpb < 4 ? 4 : (rpb > 32 ? 32 : rpb);
        const dim3 block(kWave, 4), grid((uint32_t)blocks_x, cdiv(L.oh, (uint32_t)rpb), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
#define NUS_RR(E, V, S, U)                                                                                             \
    hipLaunchKernelGGL((k_resize_rows<E, V, S, U>), grid, block, lds, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx, T.lz_ly, \
                       T.lz_ny, T.lz_wy, T.lz_stride, L.iw, L.ow, L.oh, (uint32_t)rpb, ncols_max, ipx, opx, L.in_sel)
        if (exact) {
            if (uni == 10) NUS_RR(true, true, true, 10);
            else if (uni == 12) NUS_RR(true, true, true, 12);
            else if (vec) { if (small_taps) NUS_RR(true, true, true, 0); else NUS_RR(true, true, false, 0); }
            else { if (small_taps) NUS_RR(true, false, true, 0); else NUS_RR(true, false, false, 0); }
        } else {
            if (uni == 10) NUS_RR(false, true, true, 10);
            else if (uni == 12) NUS_RR(false, true, true, 12);
            else if (vec) { if (small_taps) NUS_RR(false, true, true, 0); else NUS_RR(false, true, false, 0); }
            else { if (small_taps) NUS_RR(false, false, true, 0); else NUS_RR(false, false, false, 0); }
        }
#undef NUS_RR
    });
}

hipError_t launch_resize_win(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max,
                             uint32_t union_taps, uint32_t outputs_per_lane)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    const int vc = ncols_max <= 128 ? 2 : (ncols_max <= 192 ? 3 : 0); // 4 columns per lane: 256 VGPRs, slower than the LDS-row kernel
    if (vc == 0 || (L.ow % 4) != 0 || (outputs_per_lane != 4 && outputs_per_lane != 2)) return hipErrorInvalidValue;
    // union-window H pass of 8 taps (two outputs per lane: their windows differ by at most one column) or 10; wider unions: the plain
    // 8-slot H pass (VGPR budget)
    // (the plain pass in place of a union was measured again in round 5: equal at x1.4 / x1.7, 14 - 40 % slower at x1.3, x2.2, x2.5)
    const int uni = (union_taps > 0 && union_taps <= 8 && outputs_per_lane == 2) ? 8 : ((union_taps > 0 && union_taps <= 10) ? 10 : 0);
    // the four waves' rows, then (3 columns per lane with the union H pass) their union weights: [tap][64] float4 per wave
    const size_t lds = ((vc == 3 && uni) ? 0 : (size_t)4 * kWinDepth * vc * kWave * sizeof(uint32_t)) + // (the row rings: see the kernel)
                       (size_t)4 * (ncols_max + kResizeSlack) * sizeof(float4) +
                       ((vc == 3 && uni) ? (size_t)4 * uni * kWave * sizeof(float4) : 0);
    const uint32_t segw = 64 * outputs_per_lane;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint64_t blocks_x = cdiv(cdiv(L.ow, segw), 4);
        uint64_t rpb = (uint64_t)L.oh * blocks_x * n / 4096; // a few thousand blocks per launch ...
        rpb = rpb < 16 ? 16 : (rpb > 64 ? 64 : rpb);          // ... each tall enough to amortise its window fill
        const dim3 block(kWave, 4), grid((uint32_t)blocks_x, cdiv(L.oh, (uint32_t)rpb), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
#define NUS_RW(E, C, U, NN)                                                                                              \
    hipLaunchKernelGGL((k_resize_win<E, C, U, NN>), grid, block, lds, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx, T.lz_ly, \
                       T.lz_wy, T.lz_stride, L.iw, L.ih, L.ow, L.oh, (uint32_t)rpb, ncols_max, ipx, opx, L.in_sel)
#define NUS_RW3(E, NN)                                                           \
    if (vc == 2) { if (uni) NUS_RW(E, 2, 10, NN); else NUS_RW(E, 2, 0, NN); }    \
    else { if (uni) NUS_RW(E, 3, 10, NN); else NUS_RW(E, 3, 0, NN); }
#define NUS_RW2(E)                                                                         \
    if (outputs_per_lane == 4) { NUS_RW3(E, 4) }                                           \
    else if (uni == 8) { if (vc == 2) NUS_RW(E, 2, 8, 2); else NUS_RW(E, 3, 8, 2); }       \
    else { NUS_RW3(E, 2) }
        if (exact) {
            NUS_RW2(true)
        } else {
            NUS_RW2(false)
        }
#undef NUS_RW2
#undef NUS_RW3
#undef NUS_RW
    });
}

} // namespace nus

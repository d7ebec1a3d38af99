// nus_k_lanczos_xs.hip -- separable resize (Lanczos-3, Catmull-Rom, Triangle) at the integer factors
// x3 and x4 (720p -> 4K, 540p -> 4K): the register-window design of nus_k_lanczos_x2.hip with S output
// rows per input row and S horizontal phases per input column.
// image-0.24.9 imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251 (vertical pass
// into f32, then horizontal pass).
//
// At an integer factor S the output o = S k + p (phase p) has its taps inside a 6-slot frame that starts
// at input index k - 3 + delta_p, delta_p in {0, 1} (host-checked for every output, border windows
// included: slots that fall outside the image carry weight 0).  So exactly as at x2: phases with
// delta = 0 read the six window rows r-3 .. r+2, row r+3 then replaces row r-3, and phases with
// delta = 1 read r-2 .. r+3; horizontally a lane's 4 columns plus 3 from each neighbour lane (DPP)
// cover every frame.  The 4 S left-most and right-most output columns (border-renormalised weights)
// are left to k_lanczos_general.
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 2 // nt: see nus_k_lanczos_x2.hip; this kernel -8 ... -17 % (profiles/r04_nt_stores_by_kernel.txt)
#endif
#include "nus_device.hpp"

namespace nus {

namespace {

constexpr int kXsMaxS = 4;

// delta_p: the frame of phase p starts at k - 2 instead of k - 3 when the output centre lies right of
// input pixel k (the same rule the host uses to build and check the frames: xs_phase_delta in nus_tables.cpp)
constexpr bool xs_delta(int S, int p) { return 2 * p + 1 > S; }

struct LanczosXsArgs {
    const uint8_t *in;
    uint8_t *out;
    const float *wy6;      // [oh][6] vertical weights in the phase frame of each output row
    float w[kXsMaxS][6];   // interior weights of phase p (same numbers on both axes, host-checked); CLS: unused
    // CLS (x3): the interior weights depend on the binade of the coordinate (nus_tables.hpp): class of every input
    // column / row, and the weights of each class [class][S][6]
    const uint32_t *cls_x, *cls_y;
    const float *wcls_x, *wcls_y;
    uint32_t sel;          // input channel order
    uint32_t iw, ih;
    uint32_t nstrips, nrowblocks, th;
    size_t in_frame_bytes, out_frame_bytes;
};


// Row prefetches: as in nus_k_lanczos_x2.hip / nus_k_lanczos_r32.hip the rows are requested with LDS-DMA loads issued from
// inline assembly (16 B per lane straight into a per-wave 1-KiB LDS slot, invisible to the compiler's s_waitcnt insertion)
// and waited for with hand-counted `s_waitcnt vmcnt(N)`; tools/check_hidden_loads.py verifies the counts on the generated
// code (tests/test_kernel_asm.py).
#ifndef NUS_XS_DEPTH
#define NUS_XS_DEPTH 2 // prefetch distance in steps (a step = one input row = one request, S x S stores)
#endif
#ifndef NUS_XS_WAIT_EARLY
#define NUS_XS_WAIT_EARLY 1 // 1: wait + LDS read at the start of the step; 0: where the row is converted
#endif
#ifndef NUS_XS_PIN_FMA
#define NUS_XS_PIN_FMA 1 // FMA mode: the fused operations pinned in program order
#endif
#ifndef NUS_XS_S4_WAVES
#define NUS_XS_S4_WAVES 2 // waves per SIMD the x4 FMA-mode kernel is compiled for: 185 VGPRs.  (3: 168 and 13 spilled dwords, whose
                          // reloads bring vmcnt waits -- drains of the wave's stores -- back into the loop: 10.1 - 11.5 against 8.7 - 9.7 us
                          // per 540p -> 4K frame, profiles/r03_lanczos_xs_rework_ab.txt)
#endif
constexpr int kXsDepth = NUS_XS_DEPTH;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in this kernel uses it
__device__ __forceinline__ void xs_dma_row16(const uint8_t *base, uint32_t off, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// at most N vector memory instructions outstanding; BACK (for the checker): the BACK-th most recent request has landed
template <int N, int BACK>
__device__ __forceinline__ void xs_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}

struct XsRing {
    const uint8_t *base; // this wave's kXsDepth slots of one row (64 lanes x 16 B) as a generic pointer (reads)
    uint32_t lds;        // their byte offset in LDS (wave-uniform; requests)
    int lane;
};

__device__ __forceinline__ float xs_vgpr(float s)
{
    asm volatile("" : "+v"(s));
    return s;
}

template <bool EXACT>
__device__ __forceinline__ float xs_mac(float acc, float v, float w)
{
    if (EXACT) return mac<true>(acc, v, w); // (this kernel is faster with the compiler's batches of products, nus_device.hpp)
    return mac_tight<false, NUS_XS_PIN_FMA != 0>(acc, v, w);
}

__device__ __forceinline__ void cvt_row(const uint4 raw, float (&dst)[16])
{
    const uint32_t px[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[m * 4 + c] = ch_f32(px[m], c);
}

// 1 when every pixel of this input row held by the wave is opaque (cf. row_is_opaque in nus_k_lanczos_x2.hip)
__device__ __forceinline__ uint32_t xs_row_is_opaque(const uint4 px)
{
    const bool lane_opaque = (px.x & px.y & px.z & px.w) >= 0xFF000000u;
    return __builtin_amdgcn_ballot_w64(!lane_opaque) == 0ull ? 1u : 0u;
}

// Where a wave's output row goes: as in the x2 kernel (RowStore there) the row's S KiB are turned round in LDS --
// every lane writes the 16 S bytes it computed at 16 S * lane, then reads 16 B at 1024 q + 16 * lane for store q --
// so that each store instruction writes one contiguous KiB instead of a 16-B piece of every 16 S bytes
// (tools/probe_rw_mix.hip: partly written lines cost a write-heavy stream with reads in it a third of its rate).
template <int S>
struct XsStore {
    uint4 *stage;    // this wave's S KiB of LDS
    uint32_t off[S]; // byte offset of this lane's 16 B inside an output row, per store; 2^31 = dropped by the range check
    int lane;
};

// One output row: per channel the vertical pass of the lane's 4 columns (6 taps, window rows 0 .. 5), the lane exchange
// (3 columns from each neighbour) and the horizontal pass of the lane's 4 S output pixels (output S m + p reads frame
// columns m + delta_p .. m + delta_p + 5 of e[], e[3] is the lane's own first column), convert + pack; then the row's turn
// through LDS and its S stores.  Channel by channel so that only 4 vertical sums are live.
template <bool EXACT, int S>
__device__ __forceinline__ void xs_row(const float (&win)[6][16], const float (&wv)[6], const float (&W)[S][6],
                                       __amdgpu_buffer_rsrc_t rs, const XsStore<S> &st, uint32_t row_off, bool skip_alpha)
{
    // skip_alpha (FMA mode, wave-uniform): the six tap rows are opaque in this wave, so alpha is the constant
    // 255 (see row_is_opaque in nus_k_lanczos_x2.hip); v_cvt_pk_u8_f32 only ever replaces bytes 0..2 then
    uint32_t o[4 * S];
#pragma unroll
    for (int i = 0; i < 4 * S; ++i) o[i] = skip_alpha ? 0xFF000000u : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 3 && skip_alpha) continue;
        float v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float acc = win[0][m * 4 + c] * wv[0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
            for (int j = 1; j < 6; ++j) acc = xs_mac<EXACT>(acc, win[j][m * 4 + c], wv[j]);
            v[m] = acc;
        }
        float e[10]; // vertical sums of input columns c0-3 .. c0+6 for this channel
        e[0] = wave_up(v[1]);
        e[1] = wave_up(v[2]);
        e[2] = wave_up(v[3]);
        e[3] = v[0];
        e[4] = v[1];
        e[5] = v[2];
        e[6] = v[3];
        e[7] = wave_down(v[0]);
        e[8] = wave_down(v[1]);
        e[9] = wave_down(v[2]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int p = 0; p < S; ++p) {
                float a = e[m + (xs_delta(S, p) ? 1 : 0)] * W[p][0];
#pragma unroll
                for (int j = 1; j < 6; ++j) a = xs_mac<EXACT>(a, e[m + (xs_delta(S, p) ? 1 : 0) + j], W[p][j]);
                o[S * m + p] = pack_u8<EXACT>(a, c, o[S * m + p]);
            }
        }
    }
    // range-checked buffer stores: pieces that must not be written sit beyond num_records (see the x2 kernel), so all S
    // stores issue on every path and for every lane -- the hand-counted waits rely on exactly S per output row
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < S; ++q) st.stage[S * st.lane + q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    __builtin_amdgcn_wave_barrier(); // compiler only: same wave, LDS instructions execute in order
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const uint4 t = st.stage[64 * q + st.lane];
        const u32x4 v = {t.x, t.y, t.z, t.w};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, row_off + st.off[q], 0, NUS_STORE_AUX);
    }
    __builtin_amdgcn_wave_barrier();
}

// CLS: vertical weights of the current row's class, S x 6 numbers that change a few times per frame, kept in SGPRs; each
// output row copies its six into VGPRs (a scalar operand halves an FMA's issue rate).  Without classes the interior vertical
// weights are the horizontal ones (W, already in VGPRs).
template <int S>
struct XsRowWeights {
    float w[S][6];
    uint32_t cls;
};

// One input row r -> output rows S r .. S r + S - 1.  At entry window row j holds input row r-3+j and ring slot `pos` holds
// row r+3 (requested kXsDepth steps ago).  Unlike the x2 kernel the window is shifted, not rotated: one copy of the step's
// code instead of six (at S = 4 six copies are ~16k instructions, more than the instruction cache), for 80 register moves
// per step that is 3 - 5 % of its work.
//
// Vector memory instructions of a step, in issue order and on every path: n0 S stores (the n0 = 2 phases whose frame is
// rows r-3 .. r+2), the request of row r+3+D, (S - n0) S stores (the phases whose frame is rows r-2 .. r+3).  Issued since the
// request of row r+3 when the wave waits for it where it is converted: the rest of that step ((S - n0) S), D-1 whole
// steps (S S + 1 each), this step's first stores (n0 S): N = D (S S + 1) - 1; at the START of the step: N - n0 S.
template <bool EXACT, int S, bool CLS>
__device__ __forceinline__ void xs_step(float (&win)[6][16], const XsRing &ring, uint32_t &pos, int r, uint32_t in_off,
                                        const XsStore<S> &st, const LanczosXsArgs &A, const float (&W)[S][6], XsRowWeights<S> &RW,
                                        const uint8_t *src, __amdgpu_buffer_rsrc_t rs, uint32_t &opaque)
{
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int D = kXsDepth, N0 = 2;
    static_assert(xs_delta(S, 0) == false && xs_delta(S, 1) == false && xs_delta(S, 2) == true, "two phases before the window moves");
    constexpr bool EARLY = NUS_XS_WAIT_EARLY != 0;
    if (CLS) {
        // vertical weights of this row's class, reloaded when the class changes (a scalar load: a vector load here would make
        // the compiler wait for vmcnt(0) -- every store of the wave -- once per step)
        typedef const __attribute__((address_space(4))) uint32_t *cu32_p;
        const uint32_t cy = ((cu32_p)(uintptr_t)A.cls_y)[r < 0 ? 0 : r];
        if (cy != RW.cls) { // wave-uniform
            RW.cls = cy;
            cfloat_p wt = (cfloat_p)(uintptr_t)(A.wcls_y + (size_t)cy * (S * 6));
#pragma unroll
            for (int p = 0; p < S; ++p)
#pragma unroll
                for (int j = 0; j < 6; ++j) RW.w[p][j] = wt[p * 6 + j];
        }
    }
    const uint32_t row_bytes = A.iw * 4 * S; // one output row
    const uint32_t off0 = (uint32_t)(S * r) * row_bytes;
    const bool interior = r >= 4 && r + 5 <= (int)A.ih; // wave-uniform
    u32x4 next = {0u, 0u, 0u, 0u};
    if (EARLY) {
        xs_wait_vmcnt<D * (S * S + 1) - 1 - N0 * S, D>();
        next = *reinterpret_cast<const u32x4 *>(ring.base + pos + 16 * ring.lane);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // half 0: phases whose frame is rows r-3 .. r+2; half 1 (after the shift): rows r-2 .. r+3
#pragma unroll
        for (int p = 0; p < S; ++p) {
            if (xs_delta(S, p) != (half == 1)) continue;
            const bool skip_alpha = !EXACT && (opaque & 0x3Fu) == 0x3Fu; // bit j: window row 5-j is opaque
            // vertical weights of this output row: the interior ones or, next to the top / bottom border where the window is cut
            // and renormalised, the row's own from the table; one copy of the row's code either way, its stores on every path
            float wv[6];
            if (CLS) {
                float ws[6];
                if (interior) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) ws[j] = RW.w[p][j];
                } else {
                    cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane((uint32_t)(S * r + p)) * 6);
#pragma unroll
                    for (int j = 0; j < 6; ++j) ws[j] = wt[j];
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) wv[j] = xs_vgpr(ws[j]);
            } else {
                if (interior) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) wv[j] = W[p][j];
                } else {
                    cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane((uint32_t)(S * r + p)) * 6);
#pragma unroll
                    for (int j = 0; j < 6; ++j) wv[j] = xs_vgpr(wt[j]);
                }
            }
            xs_row<EXACT, S>(win, wv, W, rs, st, off0 + (uint32_t)p * row_bytes, skip_alpha);
        }
        if (half == 0) {
            // row r-3 out, row r+3 in; then request row r+3+D into the same ring slot
            if (!EARLY) {
                xs_wait_vmcnt<D * (S * S + 1) - 1, D>();
                next = *reinterpret_cast<const u32x4 *>(ring.base + pos + 16 * ring.lane);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) win[j][k] = win[j + 1][k];
            {
                const uint4 px = swz4(make_uint4(next.x, next.y, next.z, next.w), A.sel);
                if (!EXACT) opaque = (opaque << 1) | xs_row_is_opaque(px);
                cvt_row(px, win[5]);
            }
            int rn = r + 3 + D;
            rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
            // the slot is requested again only when its read has RETURNED (the converted row is an operand of this empty
            // statement): nothing orders a queued ds_read behind a later LDS-DMA write (see the x2 kernel)
            asm volatile("" : : "v"(win[5][0]), "v"(win[5][15]) : "memory");
            xs_dma_row16(src, in_off + (uint32_t)rn * (A.iw * 4), ring.lds + pos);
        }
    }
    pos = pos + 1024u == (uint32_t)D * 1024u ? 0u : pos + 1024u;
}

// One wave loads a strip of 256 input columns (4 per lane; lanes 1 .. 60 produce the strip's 240 input = 240 S output
// columns, lanes 0 and 61 are their halo) and walks `th` input rows with a 6-row f32 window.
template <bool EXACT, int S, bool CLS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EXACT ? 1 : (S == 4 ? NUS_XS_S4_WAVES : 3)))) void k_lanczos3_xs(const LanczosXsArgs A)
{
    const int lane = threadIdx.x & (kWave - 1);
    // each XCD gets a contiguous run of (frame, row block, strips), as in the x2 kernel
    const uint32_t vid = xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t frame = vid / gridDim.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((vid % gridDim.x) * 4 + (threadIdx.x >> 6));
    if (wave >= A.nstrips * A.nrowblocks) return;
    const uint32_t strip = wave % A.nstrips;
    const uint32_t rb = wave / A.nstrips;
    const int c = (int)(strip * kLanczosX2StripCols) - 4 + lane * 4; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - 4 ? (int)A.iw - 4 : cl;
    const uint8_t *src = A.in + (size_t)frame * A.in_frame_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)frame * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
    // lane L computes the 4 S output pixels of input columns c .. c+3 and they are stored unless it is a halo lane or its
    // columns are the edge kernel's; after the turn in LDS this lane holds, for store q, 16 B computed by lane (64 q + lane) / S
    auto computes_stored_pixels = [&](int L) {
        const int cc = (int)(strip * kLanczosX2StripCols) - 4 + L * 4;
        return L >= 1 && L <= (int)(kLanczosX2StripCols / 4) && cc >= 4 && cc + 8 <= (int)A.iw;
    };
    // (the rings come first in the block's LDS: an LDS-DMA slot address is M0 + a 12-bit instruction offset of 0)
    __shared__ uint4 lds_rows[4][kXsDepth][64];
    __shared__ uint4 lds_stage[4][64 * S];
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    XsRing ring;
    ring.base = reinterpret_cast<const uint8_t *>(&lds_rows[w][0][0]);
    ring.lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&lds_rows[w][0][0]);
    ring.lane = lane;
    XsStore<S> st;
    st.stage = lds_stage[w];
    st.lane = lane;
    {
        const int span0 = ((int)(strip * kLanczosX2StripCols) - 4) * 4 * S; // byte offset of the wave's span in an output row
#pragma unroll
        for (int q = 0; q < S; ++q)
            st.off[q] = computes_stored_pixels((64 * q + lane) / S) ? (uint32_t)(span0 + 1024 * q + 16 * lane) : 0x80000000u;
    }
    const uint32_t in_off = (uint32_t)cl * 4u; // the lane's byte offset inside an input row
    const int r0 = (int)(rb * A.th);
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    auto row_off = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return in_off + (uint32_t)rr * (A.iw * 4);
    };

    float W[S][6];
    XsRowWeights<S> RW;
    RW.cls = 0xffffffffu;
    {
        // CLS: the lane's 4 columns share a class (host-checked); lanes that do not store take class 0
        const uint32_t cx = CLS && c >= 4 && c + 8 <= (int)A.iw ? A.cls_x[c] : 0u;
        float w0[S * 6];
#pragma unroll
        for (int i = 0; i < S * 6; ++i) w0[i] = CLS ? A.wcls_x[(size_t)cx * (S * 6) + i] : A.w[i / 6][i % 6]; // all in flight together
#pragma unroll
        for (int p = 0; p < S; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                W[p][j] = xs_vgpr(w0[p * 6 + j]); // VGPR copy: scalar operands halve the VALU issue rate
                RW.w[p][j] = 0.0f;
            }
    }
    float win[6][16];
    uint32_t opaque = 0;
    {
        // the six rows of the first window (ordinary loads, all in flight together), then the first requests
        uint4 first[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) first[j] = *reinterpret_cast<const uint4 *>(src + row_off(r0 - 3 + j));
#pragma unroll
        for (int j = 0; j < kXsDepth; ++j) xs_dma_row16(src, row_off(r0 + 3 + j), ring.lds + (uint32_t)j * 1024u);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint4 px = swz4(first[j], A.sel);
            if (!EXACT) opaque = (opaque << 1) | xs_row_is_opaque(px);
            cvt_row(px, win[j]);
        }
        // the hand-counted waits of the loop assume that nothing older than its own instructions is outstanding
        xs_wait_vmcnt<0, 0>();
    }
    uint32_t pos = 0;
    for (int r = r0; r < r_end; ++r) xs_step<EXACT, S, CLS>(win, ring, pos, r, in_off, st, A, W, RW, src, rs, opaque);
}

// The 4 S left-most and right-most output columns (tap windows cut by the image border, weights
// renormalised).  As in the x2 edge kernel lanes map to input ROWS: each lane produces the 4 S x S output
// pixels of its row from a 7-row x 8-column input patch, so the horizontal weights are wave-uniform
// (kernel arguments) and the vertical ones per lane.
struct LanczosXsEdgeArgs {
    const uint8_t *in;
    uint8_t *out;
    const float *wy6;
    float wx[2][4 * kXsMaxS][6]; // [side][output column of that side][frame slot], 0 outside the image
    uint32_t sel;
    uint32_t iw, ih;
    size_t in_frame_bytes, out_frame_bytes;
};

__device__ __forceinline__ uint32_t px_of(const uint4 (&row)[2], int col)
{
    const uint4 &v = row[col >> 2];
    switch (col & 3) {
    case 0: return v.x;
    case 1: return v.y;
    case 2: return v.z;
    default: return v.w;
    }
}

template <bool EXACT, int S, int SIDE>
__device__ __forceinline__ void xs_edge_rows(const LanczosXsEdgeArgs &A, const uint4 (&raw)[7][2], int r, uint32_t *dst_frame)
{
    const uint32_t ow = A.iw * S;
#pragma unroll
    for (int p = 0; p < S; ++p) {
        const int d = xs_delta(S, p) ? 1 : 0; // rows r-3+d .. r+2+d of the patch
        const uint32_t oy = (uint32_t)(S * r + p);
        const float *wvp = A.wy6 + (size_t)oy * 6;
        float wv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wv[j] = wvp[j];
        float V[8][4];
#pragma unroll
        for (int col = 0; col < 8; ++col)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = ch_f32(px_of(raw[d], col), c) * wv[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, ch_f32(px_of(raw[d + j], col), c), wv[j]);
                V[col][c] = acc;
            }
        uint32_t o[4 * S];
#pragma unroll
        for (int q = 0; q < 4 * S; ++q) {
            // patch-local column of frame slot 0: left side k - 3 + delta with k = q / S;
            // right side (patch starts at iw - 8, outputs start at k = iw - 4): 1 + q / S + delta
            const int dq = xs_delta(S, q % S) ? 1 : 0;
            const int l0 = SIDE == 0 ? q / S - 3 + dq : 1 + q / S + dq;
            uint32_t px = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    int li = l0 + j;
                    li = li < 0 ? 0 : (li > 7 ? 7 : li); // slots outside the image carry weight 0
                    const float w = A.wx[SIDE][q][j];
                    acc = j == 0 ? V[li][c] * w : mac<EXACT>(acc, V[li][c], w);
                }
                px = pack_u8<EXACT>(acc, c, px);
            }
            o[q] = px;
        }
        uint32_t *d4 = dst_frame + (size_t)oy * ow + (SIDE == 0 ? 0 : ow - 4 * S);
#pragma unroll
        for (int q = 0; q < S; ++q) *reinterpret_cast<uint4 *>(d4 + 4 * q) = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
}

template <bool EXACT, int S>
__global__ __launch_bounds__(64) void k_lanczos3_xs_edges(const LanczosXsEdgeArgs A)
{
    const int r = (int)(blockIdx.x * kWave + threadIdx.x);
    if (r >= (int)A.ih) return;
    const int side = blockIdx.y; // 0: left, 1: right (wave-uniform)
    const int col0 = side ? (int)A.iw - 8 : 0;
    const int rmax = (int)A.ih - 1;
    const uint8_t *src = A.in + (size_t)blockIdx.z * A.in_frame_bytes;
    uint32_t *dst = reinterpret_cast<uint32_t *>(A.out + (size_t)blockIdx.z * A.out_frame_bytes);
    uint4 raw[7][2];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int rr = r - 3 + j;
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        const size_t off = ((size_t)rr * A.iw + col0) * 4;
        raw[j][0] = swz4(*reinterpret_cast<const uint4 *>(src + off), A.sel);
        raw[j][1] = swz4(*reinterpret_cast<const uint4 *>(src + off + 16), A.sel);
    }
    if (side == 0)
        xs_edge_rows<EXACT, S, 0>(A, raw, r, dst);
    else
        xs_edge_rows<EXACT, S, 1>(A, raw, r, dst);
}

} // namespace

hipError_t launch_lanczos_xs(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t factor,
                             uint32_t rows_per_wave)
{
    if (factor != 3 && factor != 4) return hipErrorInvalidValue;
    LanczosXsArgs A;
    A.wy6 = T.lz_wy6;
    for (uint32_t p = 0; p < factor; ++p)
        for (int j = 0; j < 6; ++j) A.w[p][j] = T.lz_wxs[p][j];
    A.cls_x = T.lz_xs_cls_x;
    A.cls_y = T.lz_xs_cls_y;
    A.wcls_x = T.lz_xs_wcls_x;
    A.wcls_y = T.lz_xs_wcls_y;
    const bool cls = A.cls_x != nullptr;
    if (cls && factor != 3) return hipErrorInvalidValue; // only x3 is instantiated with weight classes
    A.sel = L.in_sel;
    A.iw = L.iw;
    A.ih = L.ih;
    A.nstrips = cdiv(L.iw, kLanczosX2StripCols);
    A.th = rows_per_wave ? rows_per_wave : 24;
    A.nrowblocks = cdiv(L.ih, A.th);
    A.in_frame_bytes = (size_t)L.iw * L.ih * 4;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    const uint32_t nwaves = A.nstrips * A.nrowblocks;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
#define NUS_XS(E, F, C) hipLaunchKernelGGL((k_lanczos3_xs<E, F, C>), grid, block, 0, L.stream, A)
        if (exact) {
            if (factor == 4) NUS_XS(true, 4, false); else if (cls) NUS_XS(true, 3, true); else NUS_XS(true, 3, false);
        } else {
            if (factor == 4) NUS_XS(false, 4, false); else if (cls) NUS_XS(false, 3, true); else NUS_XS(false, 3, false);
        }
#undef NUS_XS
    });
}

hipError_t launch_lanczos_xs_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t factor)
{
    if (factor != 3 && factor != 4) return hipErrorInvalidValue;
    LanczosXsEdgeArgs A;
    A.wy6 = T.lz_wy6;
    for (uint32_t q = 0; q < 4 * factor; ++q)
        for (int j = 0; j < 6; ++j) {
            A.wx[0][q][j] = T.lz_wxs_left[q][j];
            A.wx[1][q][j] = T.lz_wxs_right[q][j];
        }
    A.sel = L.in_sel;
    A.iw = L.iw;
    A.ih = L.ih;
    A.in_frame_bytes = (size_t)L.iw * L.ih * 4;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.out = out;
        const dim3 block(kWave), grid(cdiv(L.ih, kWave), 2, n);
        if (exact) {
            if (factor == 3)
                hipLaunchKernelGGL((k_lanczos3_xs_edges<true, 3>), grid, block, 0, L.stream, A);
            else
                hipLaunchKernelGGL((k_lanczos3_xs_edges<true, 4>), grid, block, 0, L.stream, A);
        } else {
            if (factor == 3)
                hipLaunchKernelGGL((k_lanczos3_xs_edges<false, 3>), grid, block, 0, L.stream, A);
            else
                hipLaunchKernelGGL((k_lanczos3_xs_edges<false, 4>), grid, block, 0, L.stream, A);
        }
    });
}

} // namespace nus

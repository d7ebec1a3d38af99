// nus_k_lanczos_r32.hip -- separable resize (Lanczos-3, Catmull-Rom, Triangle) at the factor 3/2 on both axes
// (720p -> 1080p, 1440p -> 4K, 1080p -> 1620p): the register-window design of nus_k_lanczos_x2.hip /
// nus_k_lanczos_xs.hip with three output rows per PAIR of input rows and three horizontal phases per pair of
// input columns.  image-0.24.9 imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251 (vertical
// pass into f32, then horizontal pass).
//
// At 3/2 the output o = 3 g + p (phase p = 0, 1, 2) belongs to the input pair g = (2g, 2g+1): its centre lies at
// 2g - 1/6, 2g + 1/2, 2g + 7/6 and its taps inside the 6-slot frame that starts at input index 2g - 3 + p
// (host-checked for every output, border windows included: slots outside the image carry weight 0).  Vertically
// a wave therefore walks the input rows in pairs: phase 0 reads the window rows r-3 .. r+2, the window moves one
// row, phase 1 reads r-2 .. r+3, it moves again, phase 2 reads r-1 .. r+4 -- which is also the window of the
// next pair's phase 0.  Horizontally a lane's 4 columns (two pairs) give 6 outputs whose frames start 0, 1, 2, 2,
// 3, 4 columns into the 10 columns it holds after the lane exchange (its own 4 plus 3 from each neighbour).
// The weights of a phase move with the binade of the sample coordinate ((o + 0.5) * fl(2/3) is rounded in f32), so
// every lane / row pair takes the weight set of its class (nus_tables.hpp: lanczos_r32_weight_classes).
// The 12 left-most and right-most output columns (border-renormalised weights) belong to k_lanczos3_r32_edges.
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 2 // nt: see nus_k_lanczos_x2.hip; this kernel -8 ... -17 % (profiles/r04_nt_stores_by_kernel.txt)
#endif
#include "nus_device.hpp"

namespace nus {

namespace {

struct LanczosR32Args {
    const uint8_t *in;
    uint8_t *out;
    const float *wy6;               // [oh][6] vertical weights in the phase frame of each output row
    const uint32_t *cls_x, *cls_y;  // weight class of every input column pair / row pair
    const float *wcls_x, *wcls_y;   // [class][3][6]
    uint32_t sel;                   // input channel order
    uint32_t iw, ih;
    uint32_t nstrips, nrowblocks, th; // th: input rows per wave, even
    size_t in_frame_bytes, out_frame_bytes;
};


// Row prefetches: as in nus_k_lanczos_x2.hip the rows are requested with LDS-DMA loads issued from inline assembly (16 B per
// lane straight into a per-wave 1-KiB LDS slot, no VGPR destination, invisible to the compiler's s_waitcnt insertion) and
// waited for with hand-counted `s_waitcnt vmcnt(N)` -- gfx950 retires loads and stores in issue order on one counter, and the
// compiler's own waits in front of a row conversion drained all but the newest stores once per step.
// tools/check_hidden_loads.py verifies the counts on the generated code (tests/test_kernel_asm.py).
#ifndef NUS_R32_DEPTH
#define NUS_R32_DEPTH 2 // prefetch distance in steps (a step = two input rows = two requests)
#endif
#ifndef NUS_R32_WAIT_EARLY
#define NUS_R32_WAIT_EARLY 1 // 1: wait + LDS read at the start of the phase whose end the row is converted at; 0: at its end
#endif
#ifndef NUS_R32_PIN_FMA
#define NUS_R32_PIN_FMA 1 // the fused operations pinned in program order (mac_tight<.., true>): -5 % on 4-channel input, -4 % at 720p
#endif
constexpr int kR32Depth = NUS_R32_DEPTH;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in this kernel uses it
__device__ __forceinline__ void r32_dma_row16(const uint8_t *base, uint32_t off, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// at most N vector memory instructions outstanding; BACK (for the checker): the BACK-th most recent request has landed
template <int N, int BACK>
__device__ __forceinline__ void r32_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}

// LDS ring of one wave: 2 kR32Depth slots of one row (64 lanes x 16 B).  `pos` walks the slot pairs.
struct R32Ring {
    const uint8_t *base; // this wave's ring as a generic pointer (reads)
    uint32_t lds;        // its byte offset in LDS (wave-uniform; requests)
    int lane;
};

__device__ __forceinline__ float r32_vgpr(float s)
{
    asm volatile("" : "+v"(s));
    return s;
}

__device__ __forceinline__ void r32_cvt_row(const uint4 raw, float (&dst)[16])
{
    const uint32_t px[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[m * 4 + c] = ch_f32(px[m], c);
}

// 1 when every pixel of this input row held by the wave is opaque (cf. row_is_opaque in nus_k_lanczos_x2.hip)
__device__ __forceinline__ uint32_t r32_row_is_opaque(const uint4 px)
{
    const bool lane_opaque = (px.x & px.y & px.z & px.w) >= 0xFF000000u;
    return __builtin_amdgcn_ballot_w64(!lane_opaque) == 0ull ? 1u : 0u;
}

// Where a wave's output row goes: as in the x2 / xs kernels the row is turned round in LDS -- every lane writes the 24
// bytes (6 pixels) it computed at 24 * lane, then reads 16 B at 1024 q + 16 * lane for store q -- so that a store
// instruction writes contiguous bytes (tools/probe_rw_mix.hip).  The wave's span is 64 x 24 = 1536 B: one full store
// and one of 32 lanes.
struct R32Store {
    uint2 *stage;    // this wave's 1.5 KiB of LDS
    uint32_t off[2]; // byte offset of this lane's 16 B inside an output row, per store; 2^31 = dropped by the range check
    int lane;
};

// One output row: per channel the vertical pass of the lane's 4 columns (6 taps from window slots B .. B+5 mod 6), the
// lane exchange (3 columns from each neighbour) and the horizontal pass of the lane's 6 output pixels (2 input pairs x 3
// phases; output 3 m + p reads the columns e[2 m + p] .. e[2 m + p + 5], e[3] is the lane's own first column), convert +
// pack; then the row's turn through LDS and its two stores.  Channel by channel so that only 4 vertical sums are live.
template <bool EXACT, int B>
__device__ __forceinline__ void r32_row(const float (&win)[6][16], const float (&wv)[6], const float (&W)[3][6], __amdgpu_buffer_rsrc_t rs,
                                        const R32Store &st, uint32_t row_off, bool skip_alpha)
{
    // skip_alpha (FMA mode, wave-uniform): the six tap rows are opaque in this wave, so alpha is the constant
    // 255 (see row_is_opaque in nus_k_lanczos_x2.hip); v_cvt_pk_u8_f32 only ever replaces bytes 0..2 then
    uint32_t o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = skip_alpha ? 0xFF000000u : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 3 && skip_alpha) continue;
        float v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float acc = win[B % 6][m * 4 + c] * wv[0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
            for (int j = 1; j < 6; ++j) acc = mac_tight<EXACT, NUS_R32_PIN_FMA != 0>(acc, win[(B + j) % 6][m * 4 + c], wv[j]);
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
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float a = e[2 * m + p] * W[p][0];
#pragma unroll
                for (int j = 1; j < 6; ++j) a = mac_tight<EXACT, NUS_R32_PIN_FMA != 0>(a, e[2 * m + p + j], W[p][j]);
                o[3 * m + p] = pack_u8<EXACT>(a, c, o[3 * m + p]);
            }
        }
    }
    // range-checked buffer stores: pieces that must not be written sit beyond num_records (see the x2 kernel), so both
    // stores issue on every path and for every lane -- the hand-counted waits rely on exactly two per output row
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < 3; ++q) st.stage[3 * st.lane + q] = make_uint2(o[2 * q], o[2 * q + 1]);
    __builtin_amdgcn_wave_barrier(); // compiler only: same wave, LDS instructions execute in order
    const uint4 *stage4 = reinterpret_cast<const uint4 *>(st.stage);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 64 * q + st.lane;
        const uint4 t = stage4[i < 96 ? i : 95]; // (lanes 32 .. 63 of the second store hold nothing: dropped by their offset)
        const u32x4 v = {t.x, t.y, t.z, t.w};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, row_off + st.off[q], 0, NUS_STORE_AUX);
    }
    __builtin_amdgcn_wave_barrier();
}

// Vertical weights of the current row pair's class: 18 numbers that change a few times per frame, kept in SGPRs; each
// phase copies its six into VGPRs (a scalar operand halves an FMA's issue rate; all 18 resident in VGPRs: 163 instead of 145
// registers and no faster, profiles/r03_lanczos_r32_rework_ab.txt).
struct R32RowWeights {
    float w[3][6];
    uint32_t cls;
};

// One input row pair (r, r+1), r even -> output rows 3 r / 2 .. 3 r / 2 + 2.  At entry window slot (S + j) % 6 holds input
// row r-3+j and the ring's slot pair at `pos` holds rows r+3 / r+4 (requested kR32Depth steps ago).  Phase p reads the slots
// S+p .. S+p+5; row r-3+p dies with it and row r+3+p is converted into its slot: the window rotates, 3 steps unrolled.
//
// Vector memory instructions of a step, in issue order and on every path: 2 stores (phase 0), request of row r+3+2D,
// 2 stores (phase 1), request of row r+4+2D, 2 stores (phase 2).  Issued since the request of row r+3 when the wave waits
// for it at the END of phase 0: the rest of that step (5), D-1 whole steps (8 each), this step's first stores (2):
// N = 8 D - 1; waiting at the START of the phase: N = 8 D - 3.  The same two numbers hold for row r+4 around phase 1
// (2 + 8 (D - 1) + 5 and 2 + 8 (D - 1) + 3).  Either request is the 2 D-th most recent one when it is waited for.
struct R32StepCtx {
    const LanczosR32Args &A;
    const R32Ring &ring;
    const R32Store &st;
    const uint8_t *src;
    __amdgpu_buffer_rsrc_t rs;
    uint32_t in_off;
};

// phase P of a step (see r32_step): output row 3 r / 2 + P from the window slots S+P .. S+P+5, then (P < 2) row r+3+P in
template <bool EXACT, int S, int P>
__device__ __forceinline__ void r32_phase(float (&win)[6][16], const R32StepCtx &C, uint32_t pos, int r, bool interior,
                                          const float (&W)[3][6], const R32RowWeights &RW, uint32_t &opaque)
{
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int D = kR32Depth;
    constexpr bool EARLY = NUS_R32_WAIT_EARLY != 0;
    const LanczosR32Args &A = C.A;
    const uint32_t row_bytes = A.iw * 6; // one output row: 1.5 iw pixels
    const uint32_t oy = 3u * (uint32_t)(r >> 1) + (uint32_t)P;
    u32x4 next = {0u, 0u, 0u, 0u};
    const uint32_t slot = pos + (uint32_t)P * 1024u; // P < 2: the ring slot of row r+3+P
    if (P < 2 && EARLY) {
        r32_wait_vmcnt<8 * D - 3, 2 * D>();
        next = *reinterpret_cast<const u32x4 *>(C.ring.base + slot + 16 * C.ring.lane);
    }
    const bool skip_alpha = !EXACT && (opaque & 0x3Fu) == 0x3Fu; // bit j: the row j before the newest is opaque
    // vertical weights of this output row: the class's (SGPRs) or, next to the top / bottom border where the window is cut and
    // renormalised, the row's own from the table; one copy of the row's code either way, its stores on every path
    float wv[6];
    {
        float ws[6];
        if (interior) {
#pragma unroll
            for (int j = 0; j < 6; ++j) ws[j] = RW.w[P][j];
        } else {
            cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane(oy) * 6);
#pragma unroll
            for (int j = 0; j < 6; ++j) ws[j] = wt[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) wv[j] = r32_vgpr(ws[j]);
    }
    r32_row<EXACT, S + P>(win, wv, W, C.rs, C.st, oy * row_bytes, skip_alpha);
    if (P < 2) {
        // the oldest row out, row r+3+P in; then request row r+3+P+2D into the same ring slot
        if (!EARLY) {
            r32_wait_vmcnt<8 * D - 1, 2 * D>();
            next = *reinterpret_cast<const u32x4 *>(C.ring.base + slot + 16 * C.ring.lane);
        }
        {
            const uint4 px = swz4(make_uint4(next.x, next.y, next.z, next.w), A.sel);
            if (!EXACT) opaque = (opaque << 1) | r32_row_is_opaque(px);
            r32_cvt_row(px, win[(S + P) % 6]);
        }
        int rn = r + 3 + P + 2 * D;
        rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
        // the slot is requested again only when its read has RETURNED (the converted row is an operand of this empty
        // statement): nothing orders a queued ds_read behind a later LDS-DMA write (see the x2 kernel)
        asm volatile("" : : "v"(win[(S + P) % 6][0]), "v"(win[(S + P) % 6][15]) : "memory");
        r32_dma_row16(C.src, C.in_off + (uint32_t)rn * (A.iw * 4), C.ring.lds + slot);
    }
}

template <bool EXACT, int S>
__device__ __forceinline__ void r32_step(float (&win)[6][16], const R32StepCtx &C, uint32_t &pos, int r, const float (&W)[3][6],
                                         R32RowWeights &RW, uint32_t &opaque)
{
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    const LanczosR32Args &A = C.A;
    {
        // vertical weights of this row pair's class, reloaded when the class changes
        // (a scalar load: a vector load here would make the compiler wait for vmcnt(0) -- every store of the wave -- once per step)
        typedef const __attribute__((address_space(4))) uint32_t *cu32_p;
        const uint32_t cy = ((cu32_p)(uintptr_t)A.cls_y)[r >> 1];
        if (cy != RW.cls) { // wave-uniform
            RW.cls = cy;
            cfloat_p wt = (cfloat_p)(uintptr_t)(A.wcls_y + (size_t)cy * 18);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 6; ++j) RW.w[p][j] = wt[p * 6 + j];
        }
    }
    const bool interior = r >= 4 && r + 6 <= (int)A.ih; // wave-uniform: rows r-3 .. r+4 exist and the pair is not a border pair
    r32_phase<EXACT, S, 0>(win, C, pos, r, interior, W, RW, opaque);
    r32_phase<EXACT, S, 1>(win, C, pos, r, interior, W, RW, opaque);
    r32_phase<EXACT, S, 2>(win, C, pos, r, interior, W, RW, opaque);
    pos = pos + 2048u == (uint32_t)(2 * kR32Depth) * 1024u ? 0u : pos + 2048u;
}

// One wave loads a strip of 256 input columns (4 per lane; lanes 2 .. 61 produce the strip's 240 input = 360 output
// columns -- a 48-byte pair of lanes is three 16-byte store pieces, so the stored range starts and ends on an even lane --,
// lanes 1 and 62 are their halo) and walks `th` input rows with a 6-row f32 window.
constexpr int kR32StripCols = 240;

template <bool EXACT>
__global__ __launch_bounds__(256) void k_lanczos3_r32(const LanczosR32Args A)
{
    const int lane = threadIdx.x & (kWave - 1);
    // each XCD gets a contiguous run of (frame, row block, strips), as in the x2 kernel
    const uint32_t vid = xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t frame = vid / gridDim.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((vid % gridDim.x) * 4 + (threadIdx.x >> 6));
    if (wave >= A.nstrips * A.nrowblocks) return;
    const uint32_t strip = wave % A.nstrips;
    const uint32_t rb = wave / A.nstrips;
    const int c = (int)(strip * kR32StripCols) - 8 + lane * 4; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - 4 ? (int)A.iw - 4 : cl;
    const uint8_t *src = A.in + (size_t)frame * A.in_frame_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)frame * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
    // lane L computes the 6 output pixels of input columns c .. c+3; they are stored unless it is a halo lane or its
    // columns are among the 8 first / last of the image (the 12 edge output columns per side: k_lanczos3_r32_edges)
    auto computes_stored_pixels = [&](int L) {
        const int cc = (int)(strip * kR32StripCols) - 8 + L * 4;
        return L >= 2 && L < 2 + kR32StripCols / 4 && cc >= 8 && cc + 12 <= (int)A.iw;
    };
    // (the rings come first in the block's LDS: an LDS-DMA slot address is M0 + a 12-bit instruction offset of 0)
    __shared__ uint4 lds_rows[4][2 * kR32Depth][64];
    __shared__ uint2 lds_stage[4][64 * 3];
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    R32Ring ring;
    ring.base = reinterpret_cast<const uint8_t *>(&lds_rows[w][0][0]);
    ring.lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&lds_rows[w][0][0]);
    ring.lane = lane;
    R32Store st;
    st.stage = lds_stage[w];
    st.lane = lane;
    {
        // the wave's span starts at lane 0's pixels: byte 6 (c of lane 0) of an output row; a 16-byte piece at byte b of the
        // span holds pixels of lanes b / 24 and (b + 15) / 24, which are both stored or both not (even-lane boundaries)
        const int span0 = ((int)(strip * kR32StripCols) - 8) * 6;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = 1024 * q + 16 * lane;
            const bool ok = b + 16 <= 64 * 24 && computes_stored_pixels(b / 24) && computes_stored_pixels((b + 15) / 24);
            st.off[q] = ok ? (uint32_t)(span0 + b) : 0x80000000u;
        }
    }
    const uint32_t in_off = (uint32_t)cl * 4u; // the lane's byte offset inside an input row
    const int r0 = (int)(rb * A.th); // even
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    auto row_off = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return in_off + (uint32_t)rr * (A.iw * 4);
    };

    float W[3][6];
    R32RowWeights RW;
    RW.cls = 0xffffffffu;
    {
        // the lane's two column pairs share a class (host-checked); lanes that do not store take class 0
        const uint32_t cx = c >= 8 && c + 12 <= (int)A.iw ? A.cls_x[c >> 1] : 0u;
        float w18[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) w18[i] = A.wcls_x[(size_t)cx * 18 + i]; // all in flight together
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                W[p][j] = r32_vgpr(w18[p * 6 + j]); // VGPR copy: scalar operands halve the VALU issue rate
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
        for (int j = 0; j < 2 * kR32Depth; ++j) r32_dma_row16(src, row_off(r0 + 3 + j), ring.lds + (uint32_t)j * 1024u);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint4 px = swz4(first[j], A.sel);
            if (!EXACT) opaque = (opaque << 1) | r32_row_is_opaque(px);
            r32_cvt_row(px, win[j]);
        }
        // the hand-counted waits of the loop assume that nothing older than its own instructions is outstanding
        r32_wait_vmcnt<0, 0>();
    }
    uint32_t pos = 0;
    const R32StepCtx C = {A, ring, st, src, rs, in_off};
    for (int rbase = r0; rbase < r_end; rbase += 6) {
        // 3 steps unrolled so the rotating window indices are compile-time constants.  The block leaves the loop after its
        // last row pair: a step is never skipped with a later one still to run, so every path through the loop carries the
        // vector memory instructions the hand-counted waits assume.
#define NUS_R32_STEP(S) \
        r32_step<EXACT, S>(win, C, pos, rbase + S, W, RW, opaque); \
        if (S < 4 && rbase + S + 2 >= r_end) break
        NUS_R32_STEP(0);
        NUS_R32_STEP(2);
        NUS_R32_STEP(4);
#undef NUS_R32_STEP
    }
}

// The 12 left-most and right-most output columns (tap windows cut by the image border, weights renormalised).  As in
// the x2 / xs edge kernels lanes map to input ROWS -- here to row pairs: each lane produces the 12 x 3 output pixels of
// its pair from an 8-row x 12-column input patch, so the horizontal weights are wave-uniform (kernel arguments) and the
// vertical ones per lane (the table holds the border rows' renormalised weights).  Same order of operations as the main
// kernel and k_lanczos_general: vertical sums first, taps ascending.
struct LanczosR32EdgeArgs {
    const uint8_t *in;
    uint8_t *out;
    const float *wy6;
    float wx[2][12][6]; // [side][output column of that side][frame slot], 0 outside the image
    uint32_t sel;
    uint32_t iw, ih;
    size_t in_frame_bytes, out_frame_bytes;
};

__device__ __forceinline__ uint32_t r32_px_of(const uint4 (&row)[3], int col)
{
    const uint4 &v = row[col >> 2];
    switch (col & 3) {
    case 0: return v.x;
    case 1: return v.y;
    case 2: return v.z;
    default: return v.w;
    }
}

template <bool EXACT, int SIDE>
__device__ __forceinline__ void r32_edge_rows(const LanczosR32EdgeArgs &A, const uint4 (&raw)[8][3], int r, uint32_t *dst_frame)
{
    const uint32_t ow = A.iw / 2 * 3;
#pragma unroll
    for (int p = 0; p < 3; ++p) { // rows r-3+p .. r+2+p of the patch
        const uint32_t oy = 3u * (uint32_t)(r >> 1) + (uint32_t)p;
        const float *wvp = A.wy6 + (size_t)oy * 6;
        float wv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wv[j] = wvp[j];
        float V[12][4];
#pragma unroll
        for (int col = 0; col < 12; ++col)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = ch_f32(r32_px_of(raw[p], col), c) * wv[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, ch_f32(r32_px_of(raw[p + j], col), c), wv[j]);
                V[col][c] = acc;
            }
        uint32_t o[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            // patch-local column of frame slot 0: left side 2 (q / 3) - 3 + q % 3; right side (the patch starts at iw - 12, its
            // outputs at pair iw / 2 - 4): 1 + 2 (q / 3) + q % 3
            const int l0 = SIDE == 0 ? 2 * (q / 3) - 3 + q % 3 : 1 + 2 * (q / 3) + q % 3;
            uint32_t px = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    int li = l0 + j;
                    li = li < 0 ? 0 : (li > 11 ? 11 : li); // slots outside the image carry weight 0
                    const float w = A.wx[SIDE][q][j];
                    acc = j == 0 ? V[li][c] * w : mac<EXACT>(acc, V[li][c], w);
                }
                px = pack_u8<EXACT>(acc, c, px);
            }
            o[q] = px;
        }
        uint32_t *d4 = dst_frame + (size_t)oy * ow + (SIDE == 0 ? 0 : ow - 12);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4 *>(d4 + 4 * q) = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
}

template <bool EXACT>
__global__ __launch_bounds__(64) void k_lanczos3_r32_edges(const LanczosR32EdgeArgs A)
{
    const int r = 2 * (int)(blockIdx.x * kWave + threadIdx.x); // this lane's row pair
    if (r >= (int)A.ih) return;
    const int side = blockIdx.y; // 0: left, 1: right (wave-uniform)
    const int col0 = side ? (int)A.iw - 12 : 0;
    const int rmax = (int)A.ih - 1;
    const uint8_t *src = A.in + (size_t)blockIdx.z * A.in_frame_bytes;
    uint32_t *dst = reinterpret_cast<uint32_t *>(A.out + (size_t)blockIdx.z * A.out_frame_bytes);
    uint4 raw[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int rr = r - 3 + j;
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        const size_t off = ((size_t)rr * A.iw + col0) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) raw[j][k] = swz4(*reinterpret_cast<const uint4 *>(src + off + 16 * k), A.sel);
    }
    if (side == 0)
        r32_edge_rows<EXACT, 0>(A, raw, r, dst);
    else
        r32_edge_rows<EXACT, 1>(A, raw, r, dst);
}

} // namespace

hipError_t launch_lanczos_r32(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave)
{
    if (2 * (uint64_t)L.ow != 3 * (uint64_t)L.iw || 2 * (uint64_t)L.oh != 3 * (uint64_t)L.ih || (L.iw % 4) != 0 || (L.ih % 2) != 0 ||
        T.lz_xs_cls_x == nullptr)
        return hipErrorInvalidValue;
    LanczosR32Args A;
    A.wy6 = T.lz_wy6;
    A.cls_x = T.lz_xs_cls_x;
    A.cls_y = T.lz_xs_cls_y;
    A.wcls_x = T.lz_xs_wcls_x;
    A.wcls_y = T.lz_xs_wcls_y;
    A.sel = L.in_sel;
    A.iw = L.iw;
    A.ih = L.ih;
    A.nstrips = cdiv(L.iw, kR32StripCols);
    A.th = rows_per_wave ? (rows_per_wave + 1) & ~1u : 24;
    A.nrowblocks = cdiv(L.ih, A.th);
    A.in_frame_bytes = (size_t)L.iw * L.ih * 4;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    const uint32_t nwaves = A.nstrips * A.nrowblocks;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
        if (exact)
            hipLaunchKernelGGL(k_lanczos3_r32<true>, grid, block, 0, L.stream, A);
        else
            hipLaunchKernelGGL(k_lanczos3_r32<false>, grid, block, 0, L.stream, A);
    });
}

hipError_t launch_lanczos_r32_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact)
{
    LanczosR32EdgeArgs A;
    A.wy6 = T.lz_wy6;
    for (int q = 0; q < 12; ++q)
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
        const dim3 block(kWave), grid(cdiv(L.ih / 2, kWave), 2, n);
        if (exact)
            hipLaunchKernelGGL(k_lanczos3_r32_edges<true>, grid, block, 0, L.stream, A);
        else
            hipLaunchKernelGGL(k_lanczos3_r32_edges<false>, grid, block, 0, L.stream, A);
    });
}

} // namespace nus

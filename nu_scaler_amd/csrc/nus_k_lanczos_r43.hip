// nus_k_lanczos_r43.hip -- separable resize (Lanczos-3, Catmull-Rom, Triangle) at the factor 4/3 on both axes
// (1080p -> 1440p, 540p -> 720p, 4K -> 5120x2880): the register-window design of nus_k_lanczos_x2.hip with four
// output rows per GROUP of three input rows and four horizontal phases per group of three input columns.
// image-0.24.9 imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251 (vertical pass into f32, then
// horizontal pass).
//
// At 4/3 the output o = 4 g + p (phase p = 0 .. 3) belongs to the input group g = (3g, 3g+1, 3g+2): its centre lies at
// 3g - 1/8, 3g + 5/8, 3g + 11/8, 3g + 17/8 and its taps inside the 6-slot frame that starts at input index 3g - 3 + p
// (host-checked for every output, border windows included: slots outside the image carry weight 0).  The ratio 3/4
// and every sample centre (o + 0.5) * 0.75 - 0.5 are exact in f32, so all interior outputs of a phase share one set of
// weights, the same numbers on both axes (host-checked), as at x2 and x4.
// A lane owns ONE group of columns: 12 bytes in, its 4 outputs are 16 contiguous bytes out -- consecutive lanes
// store consecutive 16-byte pieces, so a store instruction writes contiguous bytes without a turn through LDS.
// Vertically a wave walks the input rows in groups of three: phase 0 reads the window rows r-3 .. r+2, the window
// moves one row, phase 1, ..., phase 3 reads r .. r+5, which is also the window of the next group's phase 0.
// The 8 left-most and right-most output columns (border-renormalised weights) belong to k_lanczos3_r43_edges.
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 2 // nt: see nus_k_lanczos_x2.hip; this kernel -8 ... -17 % (profiles/r04_nt_stores_by_kernel.txt)
#endif
#include "nus_device.hpp"

namespace nus {

namespace {

struct LanczosR43Args {
    const uint8_t *in;
    uint8_t *out;
    const float *wy6; // [oh][6] vertical weights in the phase frame of each output row
    float w[4][6];    // interior weights of phase p (the same numbers on both axes, host-checked)
    uint32_t sel;     // input channel order
    uint32_t iw, ih;
    uint32_t nstrips, nrowblocks, th; // th: input rows per wave, a multiple of 3
    size_t in_frame_bytes, out_frame_bytes;
};


struct Px3 {
    uint32_t x, y, z;
};

__device__ __forceinline__ void r43_cvt_row(const Px3 raw, float (&dst)[12])
{
    const uint32_t px[3] = {raw.x, raw.y, raw.z};
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[m * 4 + c] = ch_f32(px[m], c);
}

// 1 when every pixel of this input row held by the wave is opaque (cf. row_is_opaque in nus_k_lanczos_x2.hip)
__device__ __forceinline__ uint32_t r43_row_is_opaque(const Px3 px)
{
    const bool lane_opaque = (px.x & px.y & px.z) >= 0xFF000000u;
    return __builtin_amdgcn_ballot_w64(!lane_opaque) == 0ull ? 1u : 0u;
}

// Row prefetches: as in nus_k_lanczos_x2.hip / nus_k_lanczos_r32.hip the rows are requested with LDS-DMA loads issued from
// inline assembly (here 12 B per lane, global_load_lds_dwordx3, straight into a per-wave 1-KiB LDS slot -- the 12 bytes of
// lane l land at 16 l, the fourth dword is left alone: tools/probe_lds_dma_x3.hip --, invisible to the compiler's s_waitcnt
// insertion) and waited for with hand-counted `s_waitcnt vmcnt(N)`; tools/check_hidden_loads.py verifies
// the counts on the generated code (tests/test_kernel_asm.py).
#ifndef NUS_R43_DEPTH
#define NUS_R43_DEPTH 1 // prefetch distance in steps (a step = three input rows = three requests, four stores)
#endif
#ifndef NUS_R43_WAIT_EARLY
#define NUS_R43_WAIT_EARLY 1 // 1: wait + LDS read at the start of the phase whose end the row is converted at; 0: at its end
#endif
constexpr int kR43Depth = NUS_R43_DEPTH;
constexpr uint32_t kR43SlotBytes = 1024; // 64 lanes x 16 B (12 used)

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in this kernel uses it
__device__ __forceinline__ void r43_dma_row12(const uint8_t *base, uint32_t off, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx3 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// at most N vector memory instructions outstanding; BACK (for the checker): the BACK-th most recent request has landed
template <int N, int BACK>
__device__ __forceinline__ void r43_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}

struct R43Ring {
    const uint8_t *base; // this wave's 3 kR43Depth slots of one row (64 lanes x 16 B) as a generic pointer (reads)
    uint32_t lds;        // their byte offset in LDS (wave-uniform; requests)
    int lane;
};

// One output row: per channel the vertical pass of the lane's 3 columns (6 taps from window slots B .. B+5 mod 6), the lane
// exchange (3 columns from each neighbour) and the horizontal pass of the lane's 4 output pixels (output p reads the columns
// e[p] .. e[p + 5], e[3] is the lane's own first column), convert + pack; one 16-byte store.  Channel by channel so that only
// 3 vertical sums are live.  wv: the row's vertical weights in VGPRs.
template <bool EXACT, int B>
__device__ __forceinline__ void r43_row(const float (&win)[6][12], const float (&wv)[6], const float (&W)[4][6],
                                        __amdgpu_buffer_rsrc_t rs, uint32_t off, bool skip_alpha)
{
    // skip_alpha (FMA mode, wave-uniform): the six tap rows are opaque in this wave, so alpha is the constant
    // 255 (see row_is_opaque in nus_k_lanczos_x2.hip); v_cvt_pk_u8_f32 only ever replaces bytes 0..2 then
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = skip_alpha ? 0xFF000000u : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 3 && skip_alpha) continue;
        float v[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float acc = win[B % 6][m * 4 + c] * wv[0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
            for (int j = 1; j < 6; ++j) acc = mac_tight<EXACT, true>(acc, win[(B + j) % 6][m * 4 + c], wv[j]);
            v[m] = acc;
        }
        float e[9]; // vertical sums of input columns c0-3 .. c0+5 for this channel
        e[0] = wave_up(v[0]);
        e[1] = wave_up(v[1]);
        e[2] = wave_up(v[2]);
        e[3] = v[0];
        e[4] = v[1];
        e[5] = v[2];
        e[6] = wave_down(v[0]);
        e[7] = wave_down(v[1]);
        e[8] = wave_down(v[2]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // phase 3's frame is columns c0 .. c0+5: slots 0 .. 5 are e[3] .. e[8]
            float a = e[p] * W[p][0];
#pragma unroll
            for (int j = 1; j < 6; ++j) a = mac_tight<EXACT, true>(a, e[p + j], W[p][j]);
            o[p] = pack_u8<EXACT>(a, c, o[p]);
        }
    }
    // range-checked buffer store: a lane that must not write has its offset beyond num_records (see the x2 kernel), so the
    // store issues on every path and for every lane -- the hand-counted waits rely on exactly one per output row
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {o[0], o[1], o[2], o[3]};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, NUS_STORE_AUX);
}

__device__ __forceinline__ Px3 r43_ring_read(const R43Ring &ring, uint32_t slot)
{
    const uint4 v = *reinterpret_cast<const uint4 *>(ring.base + slot + 16 * ring.lane); // (.w: whatever the LDS held)
    return Px3{v.x, v.y, v.z};
}

struct R43StepCtx {
    const LanczosR43Args &A;
    const R43Ring &ring;
    const uint8_t *src;
    __amdgpu_buffer_rsrc_t rs;
    uint32_t in_off, lane_off;
};

// phase P of a step (see r43_step): output row 4 r / 3 + P from the window slots S+P .. S+P+5, then (P < 3) row r+3+P in
template <bool EXACT, int S, int P>
__device__ __forceinline__ void r43_phase(float (&win)[6][12], const R43StepCtx &C, uint32_t pos, int r, bool interior,
                                          const float (&W)[4][6], uint32_t &opaque)
{
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    constexpr int D = kR43Depth;
    constexpr bool EARLY = NUS_R43_WAIT_EARLY != 0;
    const LanczosR43Args &A = C.A;
    const uint32_t row_bytes = A.iw / 3 * 16; // one output row: 4 iw / 3 pixels
    const uint32_t oy = 4u * (uint32_t)(r / 3) + (uint32_t)P;
    Px3 next = {0u, 0u, 0u};
    const uint32_t slot = pos + (uint32_t)P * kR43SlotBytes; // P < 3: the ring slot of row r+3+P
    if (P < 3 && EARLY) {
        r43_wait_vmcnt<7 * D - 2, 3 * D>();
        next = r43_ring_read(C.ring, slot);
    }
    const bool skip_alpha = !EXACT && (opaque & 0x3Fu) == 0x3Fu; // bit j: the row j before the newest is opaque
    // vertical weights of this output row: the phase's interior ones (the horizontal weights, already in VGPRs) or, next to the
    // top / bottom border where the window is cut and renormalised, the row's own from the table; one copy of the row's code
    float wv[6];
    if (interior) {
#pragma unroll
        for (int j = 0; j < 6; ++j) wv[j] = W[P][j];
    } else {
        cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane(oy) * 6);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            wv[j] = wt[j];
            asm volatile("" : "+v"(wv[j]));
        }
    }
    r43_row<EXACT, S + P>(win, wv, W, C.rs, C.lane_off == 0x80000000u ? C.lane_off : oy * row_bytes + C.lane_off, skip_alpha);
    if (P < 3) {
        // the oldest row out, row r+3+P in; then request row r+3+P+3D into the same ring slot
        if (!EARLY) {
            r43_wait_vmcnt<7 * D - 1, 3 * D>();
            next = r43_ring_read(C.ring, slot);
        }
        {
            const uint4 s4 = swz4(make_uint4(next.x, next.y, next.z, 0xFF000000u), A.sel);
            const Px3 px = {s4.x, s4.y, s4.z};
            if (!EXACT) opaque = (opaque << 1) | r43_row_is_opaque(px);
            r43_cvt_row(px, win[(S + P) % 6]);
        }
        int rn = r + 3 + P + 3 * D;
        rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
        // the slot is requested again only when its read has RETURNED (the converted row is an operand of this empty
        // statement): nothing orders a queued ds_read behind a later LDS-DMA write (see the x2 kernel)
        asm volatile("" : : "v"(win[(S + P) % 6][0]), "v"(win[(S + P) % 6][11]) : "memory");
        r43_dma_row12(C.src, C.in_off + (uint32_t)rn * (A.iw * 4), C.ring.lds + slot);
    }
}

// One group of input rows (r, r+1, r+2), r a multiple of 3 -> output rows 4 r / 3 .. 4 r / 3 + 3.  At entry window slot
// (S + j) % 6 holds input row r-3+j and the ring's slot triple at `pos` holds rows r+3 .. r+5 (requested kR43Depth steps ago).
// Phase p reads the slots S+p .. S+p+5; row r-3+p dies with it and row r+3+p is converted into its slot: the window rotates,
// 2 steps unrolled (round 2 shifted it: 180 register moves per step).
//
// Vector memory instructions of a step, in issue order and on every path: store, request, store, request, store, request,
// store.  Issued since the request of row r+3+p when the wave waits for it at the END of phase p: 7 D - 1 for each p (the rest
// of that step, D-1 whole steps of 7, this step up to the wait); at the START of the phase: 7 D - 2.  The request is the
// 3 D-th most recent one then.
template <bool EXACT, int S>
__device__ __forceinline__ void r43_step(float (&win)[6][12], const R43StepCtx &C, uint32_t &pos, int r, const float (&W)[4][6],
                                         uint32_t &opaque)
{
    const bool interior = r >= 3 && r + 6 <= (int)C.A.ih; // wave-uniform: rows r-3 .. r+5 exist, none of the frames is cut
    r43_phase<EXACT, S, 0>(win, C, pos, r, interior, W, opaque);
    r43_phase<EXACT, S, 1>(win, C, pos, r, interior, W, opaque);
    r43_phase<EXACT, S, 2>(win, C, pos, r, interior, W, opaque);
    r43_phase<EXACT, S, 3>(win, C, pos, r, interior, W, opaque);
    pos = pos + 3 * kR43SlotBytes == (uint32_t)(3 * kR43Depth) * kR43SlotBytes ? 0u : pos + 3 * kR43SlotBytes;
}

// One wave loads a strip of 192 input columns (3 per lane; lanes 1 .. 62 produce the strip's 186 input = 248 output
// columns, lanes 0 and 63 are their halo) and walks `th` input rows with a 6-row f32 window.
constexpr int kR43StripCols = 186;

template <bool EXACT>
__global__ __launch_bounds__(256) void k_lanczos3_r43(const LanczosR43Args A)
{
    const int lane = threadIdx.x & (kWave - 1);
    // each XCD gets a contiguous run of (frame, row block, strips), as in the x2 kernel
    const uint32_t vid = xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t frame = vid / gridDim.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((vid % gridDim.x) * 4 + (threadIdx.x >> 6));
    if (wave >= A.nstrips * A.nrowblocks) return;
    const uint32_t strip = wave % A.nstrips;
    const uint32_t rb = wave / A.nstrips;
    const int c = (int)(strip * kR43StripCols) - 3 + lane * 3; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - 3 ? (int)A.iw - 3 : cl;
    const uint8_t *src = A.in + (size_t)frame * A.in_frame_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)frame * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
    // the lane's 4 output pixels are stored unless it is a halo lane or its group is one of the two first / last of the
    // image (the 8 edge output columns per side: k_lanczos3_r43_edges)
    const bool stores = lane >= 1 && lane <= kR43StripCols / 3 && c >= 6 && c + 9 <= (int)A.iw;
    const uint32_t lane_off = stores ? (uint32_t)(c / 3) * 16u : 0x80000000u; // byte offset of its 16 B inside an output row
    const uint32_t in_off = (uint32_t)cl * 4u;                               // and of its 12 B inside an input row
    const int r0 = (int)(rb * A.th); // a multiple of 3
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    auto row_off = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return in_off + (uint32_t)rr * (A.iw * 4);
    };
    __shared__ uint4 lds_rows[4][3 * kR43Depth][64];
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    R43Ring ring;
    ring.base = reinterpret_cast<const uint8_t *>(&lds_rows[w][0][0]);
    ring.lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&lds_rows[w][0][0]);
    ring.lane = lane;

    float W[4][6];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            W[p][j] = A.w[p][j];
            asm volatile("" : "+v"(W[p][j])); // VGPR copy: scalar operands halve the VALU issue rate
        }
    float win[6][12];
    uint32_t opaque = 0;
    {
        // the six rows of the first window (ordinary loads, all in flight together), then the first requests
        Px3 first[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) first[j] = *reinterpret_cast<const Px3 *>(src + row_off(r0 - 3 + j));
#pragma unroll
        for (int j = 0; j < 3 * kR43Depth; ++j) r43_dma_row12(src, row_off(r0 + 3 + j), ring.lds + (uint32_t)j * kR43SlotBytes);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint4 s4 = swz4(make_uint4(first[j].x, first[j].y, first[j].z, 0xFF000000u), A.sel);
            const Px3 px = {s4.x, s4.y, s4.z};
            if (!EXACT) opaque = (opaque << 1) | r43_row_is_opaque(px);
            r43_cvt_row(px, win[j]);
        }
        // the hand-counted waits of the loop assume that nothing older than its own instructions is outstanding
        r43_wait_vmcnt<0, 0>();
    }
    uint32_t pos = 0;
    const R43StepCtx C = {A, ring, src, rs, in_off, lane_off};
    for (int rbase = r0; rbase < r_end; rbase += 6) {
        // 2 steps unrolled so the rotating window indices are compile-time constants; the block leaves the loop after its last
        // row group, so every path through the loop carries the vector memory instructions the hand-counted waits assume
        r43_step<EXACT, 0>(win, C, pos, rbase, W, opaque);
        if (rbase + 3 >= r_end) break;
        r43_step<EXACT, 3>(win, C, pos, rbase + 3, W, opaque);
    }
}

// The 8 left-most and right-most output columns (tap windows cut by the image border, weights renormalised).  As in the
// other edge kernels lanes map to input ROWS -- here to row groups: each lane produces the 8 x 4 output pixels of its group
// from a 9-row x 12-column input patch, so the horizontal weights are wave-uniform (kernel arguments) and the vertical
// ones per lane (the table holds the border rows' renormalised weights).  Same order of operations as the main kernel.
struct LanczosR43EdgeArgs {
    const uint8_t *in;
    uint8_t *out;
    const float *wy6;
    float wx[2][8][6]; // [side][output column of that side][frame slot], 0 outside the image
    uint32_t sel;
    uint32_t iw, ih;
    size_t in_frame_bytes, out_frame_bytes;
};

__device__ __forceinline__ uint32_t r43_px_of(const uint4 (&row)[3], int col)
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
__device__ __forceinline__ void r43_edge_rows(const LanczosR43EdgeArgs &A, const uint4 (&raw)[9][3], int r, uint32_t *dst_frame)
{
    const uint32_t ow = A.iw / 3 * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) { // rows r-3+p .. r+2+p of the patch
        const uint32_t oy = 4u * (uint32_t)(r / 3) + (uint32_t)p;
        const float *wvp = A.wy6 + (size_t)oy * 6;
        float wv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wv[j] = wvp[j];
        float V[12][4];
#pragma unroll
        for (int col = 0; col < 12; ++col)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = ch_f32(r43_px_of(raw[p], col), c) * wv[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, ch_f32(r43_px_of(raw[p + j], col), c), wv[j]);
                V[col][c] = acc;
            }
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // patch-local column of frame slot 0: left side 3 (q / 4) - 3 + q % 4; right side (the patch starts at iw - 12, its
            // outputs at group iw / 3 - 2): 3 + 3 (q / 4) + q % 4
            const int l0 = SIDE == 0 ? 3 * (q / 4) - 3 + q % 4 : 3 + 3 * (q / 4) + q % 4;
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
        uint32_t *d4 = dst_frame + (size_t)oy * ow + (SIDE == 0 ? 0 : ow - 8);
        *reinterpret_cast<uint4 *>(d4) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(d4 + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

template <bool EXACT>
__global__ __launch_bounds__(64) void k_lanczos3_r43_edges(const LanczosR43EdgeArgs A)
{
    const int r = 3 * (int)(blockIdx.x * kWave + threadIdx.x); // this lane's row group
    if (r >= (int)A.ih) return;
    const int side = blockIdx.y; // 0: left, 1: right (wave-uniform)
    const int col0 = side ? (int)A.iw - 12 : 0;
    const int rmax = (int)A.ih - 1;
    const uint8_t *src = A.in + (size_t)blockIdx.z * A.in_frame_bytes;
    uint32_t *dst = reinterpret_cast<uint32_t *>(A.out + (size_t)blockIdx.z * A.out_frame_bytes);
    uint4 raw[9][3];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        int rr = r - 3 + j;
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        const size_t off = ((size_t)rr * A.iw + col0) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) raw[j][k] = swz4(*reinterpret_cast<const uint4 *>(src + off + 16 * k), A.sel);
    }
    if (side == 0)
        r43_edge_rows<EXACT, 0>(A, raw, r, dst);
    else
        r43_edge_rows<EXACT, 1>(A, raw, r, dst);
}

} // namespace

hipError_t launch_lanczos_r43(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave)
{
    if (3 * (uint64_t)L.ow != 4 * (uint64_t)L.iw || 3 * (uint64_t)L.oh != 4 * (uint64_t)L.ih || (L.iw % 12) != 0 || (L.ih % 3) != 0)
        return hipErrorInvalidValue;
    LanczosR43Args A;
    A.wy6 = T.lz_wy6;
    for (int p = 0; p < 4; ++p)
        for (int j = 0; j < 6; ++j) A.w[p][j] = T.lz_wxs[p][j];
    A.sel = L.in_sel;
    A.iw = L.iw;
    A.ih = L.ih;
    A.nstrips = cdiv(L.iw, kR43StripCols);
    A.th = rows_per_wave ? (rows_per_wave + 2) / 3 * 3 : 24;
    A.nrowblocks = cdiv(L.ih, A.th);
    A.in_frame_bytes = (size_t)L.iw * L.ih * 4;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    const uint32_t nwaves = A.nstrips * A.nrowblocks;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
        if (exact)
            hipLaunchKernelGGL(k_lanczos3_r43<true>, grid, block, 0, L.stream, A);
        else
            hipLaunchKernelGGL(k_lanczos3_r43<false>, grid, block, 0, L.stream, A);
    });
}

hipError_t launch_lanczos_r43_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact)
{
    LanczosR43EdgeArgs A;
    A.wy6 = T.lz_wy6;
    for (int q = 0; q < 8; ++q)
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
        const dim3 block(kWave), grid(cdiv(L.ih / 3, kWave), 2, n);
        if (exact)
            hipLaunchKernelGGL(k_lanczos3_r43_edges<true>, grid, block, 0, L.stream, A);
        else
            hipLaunchKernelGGL(k_lanczos3_r43_edges<false>, grid, block, 0, L.stream, A);
    });
}

} // namespace nus

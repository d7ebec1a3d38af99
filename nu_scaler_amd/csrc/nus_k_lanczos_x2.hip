// nus_k_lanczos_x2.hip -- exact-x2 separable resize (Lanczos-3, Catmull-Rom, Triangle): the dominant
// kernel of the 1080p -> 4K path.  image-0.24.9 imageops::resize as called at
// Nu_scale/src/upscale/common.rs:243-251 (vertical pass into f32, then horizontal pass).

// cache-policy bits of the output stores (0 = default, 2 = nt); tuning knob
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 0
#endif
#include "nus_device.hpp"

namespace nus {

namespace {

struct LanczosX2Args {
    const uint8_t *in;
    const uint8_t *in_b; // BLEND != 0: second frame of each pair
    float t;             // BLEND == 2: blend factor
    uint32_t sel;        // input channel order (kSelRGBA / kSelBGRA)
    uint8_t *out;
    const float *wy6; // [oh][6], phase frame: even row 2r taps rows r-3..r+2, odd row 2r+1 taps r-2..r+3
    float wxe[6];     // interior horizontal weights, even output 2k: columns k-3..k+2
    float wxo[6];     // odd output 2k+1: columns k-2..k+3
    uint32_t iw, ih;
    uint32_t nstrips, nrowblocks, th;
    size_t in_frame_bytes, in_b_frame_bytes, out_frame_bytes; // byte strides between consecutive frames
};

// Input rows of the x2 kernels.  BLEND 0: the frame itself.  BLEND 1 / 2: the zero-flow in-between
// frame of a pair (A, B) is formed on the fly -- trunc((1-t) a + t b) per channel, exactly the u8
// pixel k_blend_zero_flow would have stored (interpolation/mod.rs:407-411) -- so "interpolate, then
// upscale the interpolated frame" (nu_scaler_py/nu_scaler/main.py:999-1008) needs no round trip
// through HBM.  At t = 0.5 both products and the sum are exact and the truncation is a floor of a
// half-integer: one v_lerp_u8 per pixel.

template <int BLEND>
struct RowRaw {
    uint4 a, b;
};
template <>
struct RowRaw<0> {
    uint4 a;
};

template <int BLEND>
__device__ __forceinline__ RowRaw<BLEND> fetch_row(const uint8_t *pa, const uint8_t *pb, size_t off)
{
    RowRaw<BLEND> r;
    r.a = *reinterpret_cast<const uint4 *>(pa + off);
    if constexpr (BLEND != 0) r.b = *reinterpret_cast<const uint4 *>(pb + off);
    return r;
}

// (the blend is per channel, so the channel swizzle is applied once, to the blended pixel)
template <int BLEND>
__device__ __forceinline__ uint4 resolve_row(const RowRaw<BLEND> &r, float t, uint32_t sel)
{
    if constexpr (BLEND == 0) {
        return swz4(r.a, sel);
    } else if constexpr (BLEND == 1) {
        return swz4(make_uint4(__builtin_amdgcn_lerp(r.a.x, r.b.x, 0u), __builtin_amdgcn_lerp(r.a.y, r.b.y, 0u),
                               __builtin_amdgcn_lerp(r.a.z, r.b.z, 0u), __builtin_amdgcn_lerp(r.a.w, r.b.w, 0u)), sel);
    } else {
        const float nt = 1.0f - t;
        return swz4(make_uint4(blend_px(r.a.x, r.b.x, t, nt), blend_px(r.a.y, r.b.y, t, nt), blend_px(r.a.z, r.b.z, t, nt),
                               blend_px(r.a.w, r.b.w, t, nt)), sel);
    }
}

__device__ __forceinline__ float lane_up(float v) // value of lane-1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_down(float v) // value of lane+1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, true));
}

// gfx950 issues v_fma/v_mul/v_add_f32 with VGPR-only operands at ~2.4 cycles per wave64
// instruction, but ~4.4-5 cycles as soon as one operand is an SGPR (tools/probe_valu.hip).
// The wave-uniform interior filter weights are therefore copied into VGPRs once (the asm
// barrier keeps the compiler from folding them back into scalar operands).
__device__ __forceinline__ float vgpr(float s)
{
    asm volatile("" : "+v"(s));
    return s;
}

// Interior phase weights (even output: taps k-3..k+2, odd output: taps k-2..k+3).  At x2 on
// both axes the vertical and horizontal interior weights are the same 12 numbers (checked by
// the host), so one VGPR copy serves both passes.
struct PhaseWeights {
    float e[6], o[6];
};

__device__ __forceinline__ void cvt_row(const uint4 raw, float (&dst)[16])
{
    const uint32_t px[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[m * 4 + c] = ch_f32(px[m], c);
}

// 1 when every pixel of this input row held by the wave (all 64 lanes x 4 columns) is opaque.
// Rows whose whole 6-row tap window is opaque take the 3-channel path below: alpha of the output is
// then 255 on both the CPU and here -- the taps are normalised, sum(w) * 255 is within 1e-3 of 255 in
// f32 -- so it is stored as a constant and a quarter of the per-pixel arithmetic is skipped.
// Captured and rendered frames are opaque; frames with real alpha just take the 4-channel path.
#ifndef NUS_OPAQUE_PATH
#define NUS_OPAQUE_PATH 1 // dev macro: 0 builds the x2 kernel without the 3-channel path (A/B timing only)
#endif
__device__ __forceinline__ uint32_t row_is_opaque(const uint4 raw)
{
#if !NUS_OPAQUE_PATH
    return 0u;
#endif
    const bool lane_opaque = (raw.x & raw.y & raw.z & raw.w) >= 0xFF000000u;
    return __builtin_amdgcn_ballot_w64(!lane_opaque) == 0ull ? 1u : 0u;
}

// Vertical pass of one output row: 6 taps from window slots BASE .. BASE+5 (mod 6).
template <bool EXACT, int BASE, bool ALPHA>
__device__ __forceinline__ void lanczos_x2_vpass(const float (&win)[6][16], const float (&w)[6], float (&V)[16])
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (!ALPHA && (k & 3) == 3) continue; // V[alpha] is not read by the 3-channel horizontal pass
        float acc = win[BASE % 6][k] * w[0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
        for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, win[(BASE + j) % 6][k], w[j]);
        V[k] = acc;
    }
}

// Same for the few rows next to the top / bottom border, whose tap windows are cut and
// renormalised: per-row weights straight from the table (scalar operands; slow path).
template <bool EXACT, int BASE>
__device__ __forceinline__ void lanczos_x2_vpass_edge(const float (&win)[6][16], const float *__restrict__ wy6,
                                                      uint32_t oy, float (&V)[16])
{
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    cfloat_p w = (cfloat_p)(uintptr_t)(wy6 + (size_t)__builtin_amdgcn_readfirstlane(oy) * 6);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float acc = win[BASE % 6][k] * w[0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
        for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, win[(BASE + j) % 6][k], w[j]);
        V[k] = acc;
    }
}

// Horizontal pass of the lane's 8 output pixels, convert + pack, and the two 16-B stores.
template <bool EXACT, bool ALPHA>
__device__ __forceinline__ void lanczos_x2_hpass_store(const float (&V)[16], const PhaseWeights &W,
                                                       __amdgpu_buffer_rsrc_t rs, uint32_t off)
{
    constexpr uint32_t a0 = ALPHA ? 0u : 0xFF000000u; // 3-channel path: opaque output
    uint32_t o[8] = {a0, a0, a0, a0, a0, a0, a0, a0};
#pragma unroll
    for (int c = 0; c < (ALPHA ? 4 : 3); ++c) {
        float e[10]; // vertical sums of input columns c0-3 .. c0+6 for this channel
        e[0] = lane_up(V[1 * 4 + c]);
        e[1] = lane_up(V[2 * 4 + c]);
        e[2] = lane_up(V[3 * 4 + c]);
        e[3] = V[0 * 4 + c];
        e[4] = V[1 * 4 + c];
        e[5] = V[2 * 4 + c];
        e[6] = V[3 * 4 + c];
        e[7] = lane_down(V[0 * 4 + c]);
        e[8] = lane_down(V[1 * 4 + c]);
        e[9] = lane_down(V[2 * 4 + c]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float ae = e[m] * W.e[0];
            float ao = e[m + 1] * W.o[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                ae = mac<EXACT>(ae, e[m + j], W.e[j]);
                ao = mac<EXACT>(ao, e[m + 1 + j], W.o[j]);
            }
            o[2 * m] = pack_u8<EXACT>(ae, c, o[2 * m]);
            o[2 * m + 1] = pack_u8<EXACT>(ao, c, o[2 * m + 1]);
        }
    }
    // Buffer stores: lanes that must not write carry an offset beyond num_records and the
    // hardware range check drops them.  Unlike an exec-masked store behind a branch the
    // store instructions always issue, so the compiler can count them and wait for a
    // prefetched input row with vmcnt(N) instead of draining every store with vmcnt(0).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 lo = {o[0], o[1], o[2], o[3]}, hi = {o[4], o[5], o[6], o[7]};
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, NUS_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, NUS_STORE_AUX);
}

// One input row r -> output rows 2r (taps r-3..r+2) and 2r+1 (taps r-2..r+3).
// At entry window slot (S+j)%6 holds input row r-3+j, j = 0..5, and raw[S&1] holds row r+3.
// Row r-3 dies after the even phase, so row r+3 is converted into its slot BETWEEN the two
// phases: only 6 rows (96 VGPRs) are ever live, not 7.
template <bool EXACT, int BLEND, int S>
__device__ __forceinline__ void lanczos_x2_step(float (&win)[6][16], RowRaw<BLEND> (&raw)[2], uint32_t &opaque, int r, int cl,
                                                uint32_t lane_off, const LanczosX2Args &A, const PhaseWeights &W,
                                                const uint8_t *src, const uint8_t *src_b, __amdgpu_buffer_rsrc_t rs)
{
    const uint32_t row_bytes = A.iw * 8; // output row: 2*iw pixels
    const uint32_t off0 = lane_off + (uint32_t)(2 * r) * row_bytes;
    const bool interior = r >= 4 && r + 5 <= (int)A.ih; // wave-uniform
    // The 3-channel path is compiled into the plain FMA-mode kernel only: there it measures -6 % on opaque
    // frames (profiles/r01_lanczos_opaque_path_ab.txt); in the blend variants the second code path costs
    // the third wave per SIMD (175 VGPRs) and more than it saves, and EXACT is the register-hungry debug mode.
    constexpr bool OP = !EXACT && BLEND == 0;
    float V[16];
    // `opaque`: bit j = input row (newest - j) is opaque; the six newest rows are this phase's taps
    if (!interior) {
        lanczos_x2_vpass_edge<EXACT, S>(win, A.wy6, 2 * (uint32_t)r, V);
        lanczos_x2_hpass_store<EXACT, true>(V, W, rs, off0);
    } else if (OP && (opaque & 0x3Fu) == 0x3Fu) { // wave-uniform
        lanczos_x2_vpass<EXACT, S, false>(win, W.e, V);
        lanczos_x2_hpass_store<EXACT, false>(V, W, rs, off0);
    } else {
        lanczos_x2_vpass<EXACT, S, true>(win, W.e, V);
        lanczos_x2_hpass_store<EXACT, true>(V, W, rs, off0);
    }
    // row r+3 in, then request row r+5 into the same buffer (consumed two steps from now;
    // vmcnt retires in order, so that wait only sits behind stores at least a step old)
    {
        const uint4 px = resolve_row<BLEND>(raw[S & 1], A.t, A.sel);
        if (OP) opaque = (opaque << 1) | row_is_opaque(px);
        cvt_row(px, win[S % 6]);
    }
    {
        int rn = r + 5;
        rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
        raw[S & 1] = fetch_row<BLEND>(src, src_b, ((size_t)rn * A.iw + cl) * 4);
    }
    if (!interior) {
        lanczos_x2_vpass_edge<EXACT, S + 1>(win, A.wy6, 2 * (uint32_t)r + 1, V);
        lanczos_x2_hpass_store<EXACT, true>(V, W, rs, off0 + row_bytes);
    } else if (OP && (opaque & 0x3Fu) == 0x3Fu) {
        lanczos_x2_vpass<EXACT, S + 1, false>(win, W.o, V);
        lanczos_x2_hpass_store<EXACT, false>(V, W, rs, off0 + row_bytes);
    } else {
        lanczos_x2_vpass<EXACT, S + 1, true>(win, W.o, V);
        lanczos_x2_hpass_store<EXACT, true>(V, W, rs, off0 + row_bytes);
    }
}

// Exact x2 Lanczos-3.  One wave owns a strip of 256 input columns (4 per lane; lanes 0
// and 63 are halo lanes, lanes 1..62 produce 248 input = 496 output columns) and walks
// `th` input rows, keeping a 6-row f32 window of its columns in registers:
//   vertical pass  : 6 taps from the register window
//   horizontal pass: 6 taps over the lane's own 4 columns + 3 columns from each
//                    neighbouring lane, fetched with wave_shr/wave_shl DPP moves
// so every input byte is read once per strip-row-block and no LDS round trip or
// barrier is needed.  Output: 2 x 16-B stores per lane per output row (2 KiB per wave).
// The 8 left-most and right-most output columns (renormalised edge weights) are left
// to k_lanczos3_x2_edges.
template <bool EXACT, int BLEND>
__global__ __launch_bounds__(256) void k_lanczos3_x2(const LanczosX2Args A)
{
    const int lane = threadIdx.x & (kWave - 1);
    // each XCD gets a contiguous run of (frame, row block, strips): neighbouring waves' halo rows and columns in its L2
    const uint32_t vid = xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t frame = vid / gridDim.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((vid % gridDim.x) * 4 + (threadIdx.x >> 6));
    if (wave >= A.nstrips * A.nrowblocks) return;
    const uint32_t strip = wave % A.nstrips;
    const uint32_t rb = wave / A.nstrips;
    const int c = (int)(strip * kLanczosX2StripCols) - 4 + lane * 4; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - 4 ? (int)A.iw - 4 : cl;
    const bool do_store = lane >= 1 && lane <= (int)(kLanczosX2StripCols / 4) && c >= 4 && c + 8 <= (int)A.iw;
    const uint8_t *src = A.in + (size_t)frame * A.in_frame_bytes;
    const uint8_t *src_b = BLEND ? A.in_b + (size_t)frame * A.in_b_frame_bytes : src;
    // one buffer resource per output frame (< 2 GiB, checked by the host); non-storing lanes
    // sit at offset 2^31, outside num_records for every row
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)frame * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
    const uint32_t lane_off = do_store ? (uint32_t)c * 8u : 0x80000000u;
    const int r0 = (int)(rb * A.th);
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    auto load_row = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return fetch_row<BLEND>(src, src_b, ((size_t)rr * A.iw + cl) * 4);
    };

    PhaseWeights W;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        W.e[j] = vgpr(A.wxe[j]);
        W.o[j] = vgpr(A.wxo[j]);
    }
    float win[6][16];
    uint32_t opaque = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const uint4 px = resolve_row<BLEND>(load_row(r0 - 3 + j), A.t, A.sel);
        if (!EXACT && BLEND == 0) opaque = (opaque << 1) | row_is_opaque(px);
        cvt_row(px, win[j]);
    }
    RowRaw<BLEND> raw[2] = {load_row(r0 + 3), load_row(r0 + 4)};
    for (int rbase = r0; rbase < r_end; rbase += 6) {
        // 6-way unrolled so the rotating window indices are compile-time constants.
        if (rbase + 0 < r_end) lanczos_x2_step<EXACT, BLEND, 0>(win, raw, opaque, rbase + 0, cl, lane_off, A, W, src, src_b, rs);
        if (rbase + 1 < r_end) lanczos_x2_step<EXACT, BLEND, 1>(win, raw, opaque, rbase + 1, cl, lane_off, A, W, src, src_b, rs);
        if (rbase + 2 < r_end) lanczos_x2_step<EXACT, BLEND, 2>(win, raw, opaque, rbase + 2, cl, lane_off, A, W, src, src_b, rs);
        if (rbase + 3 < r_end) lanczos_x2_step<EXACT, BLEND, 3>(win, raw, opaque, rbase + 3, cl, lane_off, A, W, src, src_b, rs);
        if (rbase + 4 < r_end) lanczos_x2_step<EXACT, BLEND, 4>(win, raw, opaque, rbase + 4, cl, lane_off, A, W, src, src_b, rs);
        if (rbase + 5 < r_end) lanczos_x2_step<EXACT, BLEND, 5>(win, raw, opaque, rbase + 5, cl, lane_off, A, W, src, src_b, rs);
    }
}

// Edge columns of the exact-x2 Lanczos-3: the 8 left-most and 8 right-most output columns,
// whose tap windows are cut by the image border (weights renormalised over the taps
// that remain).  Here lanes map to input ROWS: each lane produces the 8x2 output pixels
// of its row pair from a 7-row x 8-column input patch, so the horizontal weights are
// wave-uniform (kernel arguments -> SGPRs) and the vertical weights per lane.
struct LanczosX2EdgeArgs {
    const uint8_t *in;
    const uint8_t *in_b;
    float t;
    uint32_t sel;
    uint8_t *out;
    const float *wy6;
    float wx[2][48]; // [side][output column 0..7 of that side][tap 0..5], phase frame, 0 outside the image
    uint32_t iw, ih;
    size_t in_frame_bytes, in_b_frame_bytes, out_frame_bytes;
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

template <bool EXACT, int SIDE>
__device__ __forceinline__ void lanczos_x2_edge_rows(const LanczosX2EdgeArgs &A, const uint4 (&raw)[7][2], int r,
                                                     uint32_t *dst_frame)
{
    const uint32_t ow = A.iw * 2;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
        const uint32_t oy = 2 * (uint32_t)r + phase;
        const float *wvp = A.wy6 + (size_t)oy * 6;
        float wv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wv[j] = wvp[j];
        float V[8][4];
#pragma unroll
        for (int col = 0; col < 8; ++col)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = ch_f32(px_of(raw[phase], col), c) * wv[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, ch_f32(px_of(raw[phase + j], col), c), wv[j]);
                V[col][c] = acc;
            }
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // patch-local column of tap 0: left side base = (q>>1) - 3 + (q&1);
            // right side (patch starts at iw-8, outputs start at k = iw-4): 1 + (q>>1) + (q&1)
            const int l0 = SIDE == 0 ? (q >> 1) - 3 + (q & 1) : 1 + (q >> 1) + (q & 1);
            uint32_t px = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    int li = l0 + j;
                    li = li < 0 ? 0 : (li > 7 ? 7 : li); // taps outside the image carry weight 0
                    const float w = A.wx[SIDE][q * 6 + j];
                    acc = j == 0 ? V[li][c] * w : mac<EXACT>(acc, V[li][c], w);
                }
                px = pack_u8<EXACT>(acc, c, px);
            }
            o[q] = px;
        }
        uint32_t *d = dst_frame + (size_t)oy * ow + (SIDE == 0 ? 0 : ow - 8);
        *reinterpret_cast<uint4 *>(d) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(d + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

template <bool EXACT, int BLEND>
__global__ __launch_bounds__(64) void k_lanczos3_x2_edges(const LanczosX2EdgeArgs A)
{
    const int r = (int)(blockIdx.x * kWave + threadIdx.x);
    if (r >= (int)A.ih) return;
    const int side = blockIdx.y; // 0: left, 1: right (wave-uniform)
    const int col0 = side ? (int)A.iw - 8 : 0;
    const int rmax = (int)A.ih - 1;
    const uint8_t *src = A.in + (size_t)blockIdx.z * A.in_frame_bytes;
    const uint8_t *src_b = BLEND ? A.in_b + (size_t)blockIdx.z * A.in_b_frame_bytes : src;
    uint32_t *dst = reinterpret_cast<uint32_t *>(A.out + (size_t)blockIdx.z * A.out_frame_bytes);
    uint4 raw[7][2];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int rr = r - 3 + j;
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        const size_t off = ((size_t)rr * A.iw + col0) * 4;
        raw[j][0] = resolve_row<BLEND>(fetch_row<BLEND>(src, src_b, off), A.t, A.sel);
        raw[j][1] = resolve_row<BLEND>(fetch_row<BLEND>(src, src_b, off + 16), A.t, A.sel);
    }
    if (side == 0)
        lanczos_x2_edge_rows<EXACT, 0>(A, raw, r, dst);
    else
        lanczos_x2_edge_rows<EXACT, 1>(A, raw, r, dst);
}

} // namespace

hipError_t launch_lanczos_x2(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave)
{
    LanczosX2Args A;
    A.wy6 = T.lz_wy6;
    for (int j = 0; j < 6; ++j) {
        A.wxe[j] = T.lz_wxe[j];
        A.wxo[j] = T.lz_wxo[j];
    }
    A.iw = L.iw;
    A.ih = L.ih;
    A.nstrips = cdiv(L.iw, kLanczosX2StripCols);
    A.th = rows_per_wave ? rows_per_wave : 32;
    A.nrowblocks = cdiv(L.ih, A.th);
    A.in_frame_bytes = L.in_stride ? L.in_stride : (size_t)L.iw * L.ih * 4;
    A.in_b_frame_bytes = L.in_b_stride;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    A.t = L.blend_t;
    A.sel = L.in_sel;
    const int blend = L.in_b == nullptr ? 0 : (L.blend_t == 0.5f ? 1 : 2);
    const uint32_t nwaves = A.nstrips * A.nrowblocks;
    hipError_t e = for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.in_b = L.in_b ? L.in_b + chunk_first_frame(L, in) * L.in_b_stride : nullptr;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
#define NUS_LZ(E, B) hipLaunchKernelGGL((k_lanczos3_x2<E, B>), grid, block, 0, L.stream, A)
        if (exact) {
            if (blend == 0) NUS_LZ(true, 0); else if (blend == 1) NUS_LZ(true, 1); else NUS_LZ(true, 2);
        } else {
            if (blend == 0) NUS_LZ(false, 0); else if (blend == 1) NUS_LZ(false, 1); else NUS_LZ(false, 2);
        }
#undef NUS_LZ
    });
    return e;
}

hipError_t launch_lanczos_x2_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact)
{
    LanczosX2EdgeArgs A;
    A.wy6 = T.lz_wy6;
    for (int i = 0; i < 48; ++i) {
        A.wx[0][i] = T.lz_wx_left[i];
        A.wx[1][i] = T.lz_wx_right[i];
    }
    A.iw = L.iw;
    A.ih = L.ih;
    A.in_frame_bytes = L.in_stride ? L.in_stride : (size_t)L.iw * L.ih * 4;
    A.in_b_frame_bytes = L.in_b_stride;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    A.t = L.blend_t;
    A.sel = L.in_sel;
    const int blend = L.in_b == nullptr ? 0 : (L.blend_t == 0.5f ? 1 : 2);
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.in_b = L.in_b ? L.in_b + chunk_first_frame(L, in) * L.in_b_stride : nullptr;
        A.out = out;
        const dim3 block(kWave), grid(cdiv(L.ih, kWave), 2, n);
#define NUS_LZE(E, B) hipLaunchKernelGGL((k_lanczos3_x2_edges<E, B>), grid, block, 0, L.stream, A)
        if (exact) {
            if (blend == 0) NUS_LZE(true, 0); else if (blend == 1) NUS_LZE(true, 1); else NUS_LZE(true, 2);
        } else {
            if (blend == 0) NUS_LZE(false, 0); else if (blend == 1) NUS_LZE(false, 1); else NUS_LZE(false, 2);
        }
#undef NUS_LZE
    });
}

} // namespace nus

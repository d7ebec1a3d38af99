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
#define NUS_STORE_AUX 0
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

// Vertical pass of one output row: 6 taps from the window rows 0 .. 5.  W: VGPR weights (interior rows)
// or a scalar pointer into the table (rows whose window is cut by the top / bottom border).
template <bool EXACT, typename W>
__device__ __forceinline__ void xs_vpass(const float (&win)[6][16], const W &w, float (&V)[16], bool skip_alpha)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if ((k & 3) == 3 && skip_alpha) continue; // wave-uniform; V[alpha] is then not read
        float acc = win[0][k] * w[0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
        for (int j = 1; j < 6; ++j) acc = mac<EXACT>(acc, win[j][k], w[j]);
        V[k] = acc;
    }
}

// Horizontal pass of one output row: the lane's 4 S output pixels (4 input columns x S phases),
// convert + pack, S 16-B stores.  Output pixel S m + p reads frame columns m + delta_p .. m + delta_p + 5
// of e[] (e[3] is the lane's own first column).
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

template <bool EXACT, int S>
__device__ __forceinline__ void xs_hpass_store(const float (&V)[16], const float (&W)[S][6],
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
        float e[10]; // vertical sums of input columns c0-3 .. c0+6 for this channel
        e[0] = wave_up(V[1 * 4 + c]);
        e[1] = wave_up(V[2 * 4 + c]);
        e[2] = wave_up(V[3 * 4 + c]);
        e[3] = V[0 * 4 + c];
        e[4] = V[1 * 4 + c];
        e[5] = V[2 * 4 + c];
        e[6] = V[3 * 4 + c];
        e[7] = wave_down(V[0 * 4 + c]);
        e[8] = wave_down(V[1 * 4 + c]);
        e[9] = wave_down(V[2 * 4 + c]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int p = 0; p < S; ++p) {
                float a = e[m + (xs_delta(S, p) ? 1 : 0)] * W[p][0];
#pragma unroll
                for (int j = 1; j < 6; ++j) a = mac<EXACT>(a, e[m + (xs_delta(S, p) ? 1 : 0) + j], W[p][j]);
                o[S * m + p] = pack_u8<EXACT>(a, c, o[S * m + p]);
            }
        }
    }
    // range-checked buffer stores: pieces that must not be written sit beyond num_records (see the x2 kernel)
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

// One input row r -> output rows S r .. S r + S - 1.  At entry window row j holds input row r-3+j,
// raw0 / raw1 hold rows r+3 / r+4.  Unlike the x2 kernel the window is shifted, not rotated: one copy of
// the step's code instead of six (at S = 4 six copies are ~16k instructions, more than the instruction
// cache), for 80 register moves per step that is 3 % of its work.
template <bool EXACT, int S, bool CLS>
__device__ __forceinline__ void xs_step(float (&win)[6][16], uint4 &raw0, uint4 &raw1, int r, int cl, const XsStore<S> &st,
                                        const LanczosXsArgs &A, const float (&W)[S][6], float (&Wv)[S][6], uint32_t &row_cls,
                                        const uint8_t *src, __amdgpu_buffer_rsrc_t rs, uint32_t &opaque)
{
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    if (CLS) {
        // vertical weights of this row's class: VGPR copies, reloaded when the class changes (a few times per frame)
        const uint32_t cy = __builtin_amdgcn_readfirstlane(A.cls_y[r < 0 ? 0 : r]);
        if (cy != row_cls) { // wave-uniform
            row_cls = cy;
            cfloat_p wt = (cfloat_p)(uintptr_t)(A.wcls_y + (size_t)cy * (S * 6));
#pragma unroll
            for (int p = 0; p < S; ++p)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    Wv[p][j] = wt[p * 6 + j];
                    asm volatile("" : "+v"(Wv[p][j]));
                }
        }
    }
    const uint32_t row_bytes = A.iw * 4 * S; // one output row
    const uint32_t off0 = (uint32_t)(S * r) * row_bytes;
    const bool interior = r >= 4 && r + 5 <= (int)A.ih; // wave-uniform
    float V[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // half 0: phases whose frame is rows r-3 .. r+2; half 1 (after the shift): rows r-2 .. r+3
#pragma unroll
        for (int p = 0; p < S; ++p) {
            if (xs_delta(S, p) != (half == 1)) continue;
            const bool skip_alpha = !EXACT && (opaque & 0x3Fu) == 0x3Fu; // bit j: window row 5-j is opaque
            if (interior) {
                xs_vpass<EXACT>(win, CLS ? Wv[p] : W[p], V, skip_alpha);
            } else {
                cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane((uint32_t)(S * r + p)) * 6);
                xs_vpass<EXACT>(win, wt, V, skip_alpha);
            }
            xs_hpass_store<EXACT, S>(V, W, rs, st, off0 + (uint32_t)p * row_bytes, skip_alpha);
        }
        if (half == 0) {
            // row r-3 out, row r+3 in; then request row r+5
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) win[j][k] = win[j + 1][k];
            {
                const uint4 px = swz4(raw0, A.sel);
                if (!EXACT) opaque = (opaque << 1) | xs_row_is_opaque(px);
                cvt_row(px, win[5]);
            }
            raw0 = raw1;
            int rn = r + 5;
            rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
            raw1 = *reinterpret_cast<const uint4 *>(src + ((size_t)rn * A.iw + cl) * 4);
        }
    }
}

// One wave loads a strip of 256 input columns (4 per lane; lanes 1 .. 60 produce the strip's 240 input = 240 S output
// columns, lanes 0 and 61 are their halo) and walks `th` input rows with a 6-row f32 window.
template <bool EXACT, int S, bool CLS>
__global__ __launch_bounds__(256) void k_lanczos3_xs(const LanczosXsArgs A)
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
    __shared__ uint4 lds_stage[4][64 * S];
    XsStore<S> st;
    st.stage = lds_stage[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    st.lane = lane;
    {
        const int span0 = ((int)(strip * kLanczosX2StripCols) - 4) * 4 * S; // byte offset of the wave's span in an output row
#pragma unroll
        for (int q = 0; q < S; ++q)
            st.off[q] = computes_stored_pixels((64 * q + lane) / S) ? (uint32_t)(span0 + 1024 * q + 16 * lane) : 0x80000000u;
    }
    const int r0 = (int)(rb * A.th);
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    auto load_row = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return *reinterpret_cast<const uint4 *>(src + ((size_t)rr * A.iw + cl) * 4);
    };

    float W[S][6], Wv[S][6];
    uint32_t row_cls = 0xffffffffu;
    {
        // CLS: the lane's 4 columns share a class (host-checked); lanes that do not store take class 0
        const uint32_t cx = CLS && c >= 4 && c + 8 <= (int)A.iw ? A.cls_x[c] : 0u;
#pragma unroll
        for (int p = 0; p < S; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                W[p][j] = CLS ? A.wcls_x[(size_t)cx * (S * 6) + p * 6 + j] : A.w[p][j];
                asm volatile("" : "+v"(W[p][j])); // VGPR copy: scalar operands halve the VALU issue rate
                Wv[p][j] = 0.0f;
            }
    }
    float win[6][16];
    uint32_t opaque = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const uint4 px = swz4(load_row(r0 - 3 + j), A.sel);
        if (!EXACT) opaque = (opaque << 1) | xs_row_is_opaque(px);
        cvt_row(px, win[j]);
    }
    uint4 raw0 = load_row(r0 + 3), raw1 = load_row(r0 + 4);
    for (int r = r0; r < r_end; ++r) xs_step<EXACT, S, CLS>(win, raw0, raw1, r, cl, st, A, W, Wv, row_cls, src, rs, opaque);
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

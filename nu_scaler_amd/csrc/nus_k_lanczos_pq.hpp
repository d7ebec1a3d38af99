// nus_k_lanczos_pq.hpp -- separable resize (Lanczos-3, Catmull-Rom, Triangle) at the small rational factors P/Q that have no kernel
// of their own: 5/4 (864p -> 1080p), 6/5 (900p -> 1080p), 7/5, 8/5, 9/5, 5/3 (1080p -> 1800p), 5/2 (864p -> 2160p), 7/2 -- tenths of the reference's
// scale slider (nu_scaler_py/nu_scaler/main.py:457-459) and the common capture sizes.  The register-window design of
// nus_k_lanczos_x2.hip / nus_k_lanczos_r43.hip with P output rows per GROUP of Q input rows and P horizontal phases per group of Q
// input columns.  image-0.24.9 imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251 (vertical pass into f32, then
// horizontal pass).
//
// Output o = P g + p (phase p) belongs to the input group g = (Q g .. Q g + Q - 1).  Its sample centre is (o + 0.5) Q / P - 0.5 and
// its six non-zero taps lie in the 6-slot frame that starts at input index Q g + s(p),  s(p) = floor(((2p + 1) Q - 7 P) / 2P) + 1
// (host-checked against the tables for every output, border windows included: slots outside the image carry weight 0).  Unlike
// 3/4, the ratios 4/5, 5/6, 3/5, 2/5 are NOT exact in f32: image-0.24 computes the centre as f32(o + 0.5) * f32(Q / P), whose
// rounding moves with o, so outputs of one phase do not share one set of weights (tools/: 1 - 8 distinct sets per phase).  The
// weights therefore come from the tables, in frame form: a lane keeps the 6 P horizontal weights of ITS P output columns in
// registers for the whole walk (they do not depend on the row), and every output row brings its 6 vertical weights through scalar
// loads issued one row ahead.  Same numbers, same order of operations as the general kernels: EXACT mode is bit-identical to them
// and to the oracle; the border columns and rows need no kernel of their own (their renormalised weights are table entries too).
//
// A lane owns ONE group of columns: 4 Q bytes in, P outputs; a wave's output row is turned round in LDS and stored as contiguous
// 16-byte pieces (ow % 4 == 0, host-checked; other widths keep the any-scale kernels).  Vertically a wave walks the input rows in groups of Q: phase p reads the window rows Q g + s(p) .. + 5; where s(p + 1) =
// s(p) + 1 the window moves one row between the two phases (compile-time pattern, Q moves per group).
#pragma once
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 2 // nt: see nus_k_lanczos_x2.hip
#endif
#include "nus_device.hpp"

namespace nus {

namespace {

// acc + v w in the mode's arithmetic, every operation consumed where it is formed (mac_tight pins the EXACT product only: with the
// sum pinned too the EXACT kernels need 60 - 100 registers fewer here)
template <bool EXACT>
__device__ __forceinline__ float pq_mac(float acc, float v, float w)
{
    float r = mac_tight<EXACT, true>(acc, v, w);
    if (EXACT) asm volatile("" : "+v"(r));
    return r;
}

constexpr int pq_floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <int P, int Q>
struct PqGeom {
    static_assert(P > Q && Q >= 2 && Q <= 5 && P <= 9, "up-scaling by a small rational factor");
    static constexpr int s(int p) { return pq_floordiv((2 * p + 1) * Q - 7 * P, 2 * P) + 1; } // first frame slot of phase p
    static constexpr int adv(int p) { return s(p) + 3; }                                      // window moves before phase p
    // 1: the window moves one row after phase p (for the last phase: into the next group's phase 0, s = -3 there)
    static constexpr int moves(int p) { return p + 1 < P ? s(p + 1) - s(p) : Q - 3 - s(P - 1); }
    static constexpr int R = s(P - 1) + 5;                                                    // right-most column a lane's outputs read
    static constexpr int NE = R + 4;                                                          // columns -3 .. R
    // lanes to either side a lane's outputs reach into: ceil(3 / Q) on the left, ceil((R - (Q - 1)) / Q) = R / Q on the right
    static constexpr int HL = (R / Q) > ((3 + Q - 1) / Q) ? (R / Q) : ((3 + Q - 1) / Q);
    // halo lanes per side of a strip: HL, or one more where that makes the storing lanes' 4 P-byte pieces a whole number of 16-byte
    // pieces per row (the turned stores below)
    static constexpr int HS = ((kWave - 2 * HL) * P) % 4 == 0 ? HL : HL + 1;
    static constexpr int NS = kWave - 2 * HS;                                                 // storing lanes
    static constexpr int kStripCols = NS * Q;                                                 // input columns a strip produces from
    // row requests: LDS-DMA pieces of 4 / 3 / 1 dwords (gfx950 has no 2-dword form)
    static constexpr int REQ = Q == 2 ? 2 : (Q == 5 ? 2 : 1);
    static constexpr uint32_t kSlotBytes = Q == 2 ? 512 : (Q == 5 ? 1280 : 1024);
    // stores per output row: the row's NS P dwords go through LDS and leave as SP instructions of 64 contiguous 16-byte pieces
    // (the host requires ow % 4 == 0, so that every strip's row segment is a whole number of pieces)
    static constexpr int SP = (NS * P + 255) / 256;
    static_assert((NS * P) % 4 == 0 && SP <= 3, "whole 16-byte pieces, at most three per lane");
    static constexpr int V = P * SP + Q * REQ; // vector memory instructions per step
    static constexpr int UNROLL = Q == 3 ? 2 : (Q == 5 ? 6 : 3); // steps until the 6-slot window is back at slot 0
    static constexpr bool ok()
    {
        int sum = 0;
        for (int p = 0; p < P; ++p) {
            if (moves(p) != 0 && moves(p) != 1) return false;
            sum += moves(p);
        }
        return sum == Q && s(0) == -3 && HS <= 2;
    }
    static_assert(ok(), "window pattern");
};

struct LanczosPqArgs {
    const uint8_t *in;
    uint8_t *out;
    const float *wx6; // [ow][6] horizontal weights in the phase frame of each output column
    const float *wy6; // [oh][6] vertical weights in the phase frame of each output row
    uint32_t sel;     // input channel order
    uint32_t iw, ih, oh;
    uint32_t nstrips, nrowblocks, th; // th: input rows per wave, a multiple of Q
    size_t in_frame_bytes, out_frame_bytes;
};

template <int Q>
struct PxQ {
    uint32_t v[Q];
};

template <int Q>
__device__ __forceinline__ void pq_cvt_row(const PxQ<Q> &px, float (&dst)[4 * Q])
{
#pragma unroll
    for (int m = 0; m < Q; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[m * 4 + c] = ch_f32(px.v[m], c);
}

template <int Q>
__device__ __forceinline__ PxQ<Q> pq_swz(const PxQ<Q> &px, uint32_t sel)
{
    PxQ<Q> r;
#pragma unroll
    for (int m = 0; m < Q; ++m) r.v[m] = swz(px.v[m], sel);
    return r;
}

// 1 when every pixel of this input row held by the wave is opaque (cf. row_is_opaque in nus_k_lanczos_x2.hip)
template <int Q>
__device__ __forceinline__ uint32_t pq_row_is_opaque(const PxQ<Q> &px)
{
    uint32_t a = px.v[0];
#pragma unroll
    for (int m = 1; m < Q; ++m) a &= px.v[m];
    return __builtin_amdgcn_ballot_w64(a < 0xFF000000u) == 0ull ? 1u : 0u;
}

// Row prefetches: as in nus_k_lanczos_x2.hip / _r32 / _r43 the rows are requested with LDS-DMA loads issued from inline assembly
// (invisible to the compiler's s_waitcnt insertion) and waited for with hand-counted `s_waitcnt vmcnt(N)`;
// tools/check_hidden_loads.py verifies the counts on the generated code (tests/test_kernel_asm.py).
#ifndef NUS_PQ_DEPTH
#define NUS_PQ_DEPTH 1 // prefetch distance in steps (a step = Q input rows = Q requests, P output rows)
#endif
constexpr int kPqDepth = NUS_PQ_DEPTH;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in this kernel uses it
template <int N>
__device__ __forceinline__ void pq_dma(const uint8_t *base, uint32_t off, uint32_t lds)
{
    static_assert(N == 1 || N == 3 || N == 4, "LDS-DMA widths of gfx950");
    if constexpr (N == 4)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
    else if constexpr (N == 3)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx3 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// the Q pixels of one row for every lane: into the ring slot at LDS byte offset `lds`
template <int Q>
__device__ __forceinline__ void pq_request_row(const uint8_t *base, uint32_t off, uint32_t lds)
{
    if constexpr (Q == 2) {
        pq_dma<1>(base, off, lds);            // lane l: dword at 4 l
        pq_dma<1>(base, off + 4u, lds + 256u);
    } else if constexpr (Q == 3) {
        pq_dma<3>(base, off, lds);            // lane l: 12 bytes at 16 l (tools/probe_lds_dma_x3.hip)
    } else if constexpr (Q == 4) {
        pq_dma<4>(base, off, lds);
    } else {
        pq_dma<4>(base, off, lds);
        pq_dma<1>(base, off + 16u, lds + 1024u);
    }
}

template <int Q>
__device__ __forceinline__ PxQ<Q> pq_ring_read(const uint8_t *ring, uint32_t slot, int lane)
{
    PxQ<Q> r;
    if constexpr (Q == 2) {
        r.v[0] = *reinterpret_cast<const uint32_t *>(ring + slot + 4 * lane);
        r.v[1] = *reinterpret_cast<const uint32_t *>(ring + slot + 256 + 4 * lane);
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(ring + slot + 16 * lane); // (Q == 3: .w is whatever the LDS held)
        r.v[0] = v.x, r.v[1] = v.y, r.v[2] = v.z;
        if constexpr (Q >= 4) r.v[3] = v.w;
        if constexpr (Q == 5) r.v[4] = *reinterpret_cast<const uint32_t *>(ring + slot + 1024 + 4 * lane);
    }
    return r;
}

// at most N vector memory instructions outstanding; BACK (for the checker): the BACK-th most recent request has landed
template <int N, int BACK>
__device__ __forceinline__ void pq_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}

// the value `v` holds D lanes away (D = -2 .. 2; lanes without such a neighbour get 0: they are halo lanes)
template <int D>
__device__ __forceinline__ float pq_from_lane(float v)
{
    if constexpr (D == 0) return v;
    else if constexpr (D == -1) return wave_up(v);
    else if constexpr (D == -2) return wave_up(wave_up(v));
    else if constexpr (D == 1) return wave_down(v);
    else return wave_down(wave_down(v));
}

template <int P, int Q, int K>
__device__ __forceinline__ void pq_gather(const float (&v)[Q], float (&e)[PqGeom<P, Q>::NE])
{
    if constexpr (K < PqGeom<P, Q>::NE) {
        constexpr int col = K - 3, d = pq_floordiv(col, Q), m = col - d * Q;
        e[K] = pq_from_lane<d>(v[m]);
        pq_gather<P, Q, K + 1>(v, e);
    }
}

// One output row: per channel the vertical pass of the lane's Q columns (6 taps from window slots B .. B+5 mod 6), the lane
// exchange (columns -3 .. R around the lane's first one) and the horizontal pass of the lane's P output pixels (output p reads the
// columns s(p) .. s(p) + 5), convert + pack; SP stores.  Channel by channel so that only Q vertical sums are live.
// Where a wave's output row goes (cf. RowStore in nus_k_lanczos_x2.hip: store instructions whose lanes write 16 bytes at a 4 P-byte
// stride leave every line partly written until the row's other instruction fills it in, at almost twice the cost of whole pieces --
// measured here too: each lane storing its own 4 P bytes was 13 ... 40 % slower on opaque frames).
struct PqStore {
    uint32_t *stage;      // this wave's 64 P dwords of LDS
    uint32_t widx;        // where this lane's P dwords go: storing lane k of the strip at k P, halo lanes behind them
    uint32_t off[3];      // byte offset inside an output row of the lane's 16-byte piece of store 0 .. SP - 1 (2^31: none)
    int lane;
};

// NARROW (round 6): a filter of support 2 -- Catmull-Rom, the reference's Bicubic (Nu_scale/src/upscale/common.rs:233-241) -- has its
// non-zero taps in slots 1 .. 4 of the 6-slot frame (the frame starts at the first input index inside (c - 3, c + 3), the filter's
// taps lie inside (c - 2, c + 2)): slots 0 and 5 carry weight 0 for every output, which the host checks on the tables of both axes
// before it asks for this form.  Both passes then run 4 multiply-adds per sum instead of 6.  v * 0 changes no sum (at most the sign
// of a zero, which no later operation sees), so the outputs are those of the 6-tap form bit for bit, in both modes
// (tests/test_gpu_parity.py::test_pq_narrow_taps_are_the_six_tap_kernel_bit_for_bit).
template <bool EXACT, int P, int Q, int B, bool NARROW>
__device__ __forceinline__ void pq_row(const float (&win)[6][4 * Q], const float (&wv)[6], const float (&W)[P][6],
                                       __amdgpu_buffer_rsrc_t rs, const PqStore &st, uint32_t row_off, bool skip_alpha)
{
    using G = PqGeom<P, Q>;
    constexpr int J0 = NARROW ? 1 : 0, J1 = NARROW ? 5 : 6; // taps J0 .. J1 - 1 of the frame
    uint32_t o[P];
#pragma unroll
    for (int i = 0; i < P; ++i) o[i] = skip_alpha ? 0xFF000000u : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 3 && skip_alpha) continue;
        float v[Q];
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            float acc = win[(B + J0) % 6][m * 4 + c] * wv[J0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
            for (int j = J0 + 1; j < J1; ++j) acc = pq_mac<EXACT>(acc, win[(B + j) % 6][m * 4 + c], wv[j]);
            v[m] = acc;
        }
        float e[G::NE];
        pq_gather<P, Q, 0>(v, e);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float a = e[G::s(p) + 3 + J0] * W[p][J0];
#pragma unroll
            for (int j = J0 + 1; j < J1; ++j) a = pq_mac<EXACT>(a, e[G::s(p) + 3 + j], W[p][j]);
            o[p] = pack_u8<EXACT>(a, c, o[p]);
        }
    }
    // The row through LDS (instructions of one wave execute in order there: no barrier between the lanes' writes and the reads of
    // other lanes' dwords), then range-checked buffer stores: a lane without a piece has its offset beyond num_records (see the x2
    // kernel), so the stores issue on every path and for every lane -- the hand-counted waits rely on exactly SP per output row
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t *mine = st.stage + st.widx;
#pragma unroll
    for (int i = 0; i < P; ++i) mine[i] = o[i];
    __builtin_amdgcn_wave_barrier(); // (compiler only)
    u32x4 piece[G::SP];
#pragma unroll
    for (int k = 0; k < G::SP; ++k) piece[k] = *reinterpret_cast<const u32x4 *>(st.stage + 4 * (st.lane + kWave * k));
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < G::SP; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(piece[k], rs, st.off[k] == 0x80000000u ? st.off[k] : row_off + st.off[k], 0, NUS_STORE_AUX);
}

struct PqStepCtx {
    const LanczosPqArgs &A;
    const uint8_t *ring; // this wave's Q kPqDepth row slots as a generic pointer (reads)
    uint32_t ring_lds;   // their byte offset in LDS (wave-uniform; requests)
    int lane;
    const uint8_t *src;
    __amdgpu_buffer_rsrc_t rs;
    uint32_t in_off;
    PqStore st;
};

// Phase PH of a step (see pq_step): output row P r / Q + PH from the window slots S + adv(PH) .. + 5, then -- where the window
// moves after this phase -- row r + 3 + adv(PH) in.  wnext: the NEXT output row's vertical weights (scalar registers), loaded
// during this phase.
template <bool EXACT, int P, int Q, int S, int PH, bool NARROW>
__device__ __forceinline__ void pq_phase(float (&win)[6][4 * Q], const PqStepCtx &C, uint32_t pos, int r, const float (&W)[P][6],
                                         float (&wnext)[6], uint32_t &opaque)
{
    using G = PqGeom<P, Q>;
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    constexpr int D = kPqDepth;
    constexpr bool MOVES = G::moves(PH) == 1;
    constexpr int ADV = G::adv(PH);
    const LanczosPqArgs &A = C.A;
    const uint32_t row_bytes = A.iw / Q * (4 * P); // one output row: P iw / Q pixels
    const uint32_t oy = (uint32_t)P * (uint32_t)(r / Q) + (uint32_t)PH;
    PxQ<Q> next;
    const uint32_t slot = pos + (uint32_t)ADV * G::kSlotBytes; // the ring slot of row r + 3 + ADV
    if constexpr (MOVES) {
        // Issued since the (last piece of the) request of row r + 3 + ADV, D steps ago at the end of this same phase: D steps of V
        // instructions less the REQ pieces of the request about to be made and this phase's SP stores, which follow the wait.
        // That piece is the ((Q D - 1) REQ + 1)-th most recent request.
        pq_wait_vmcnt<D * G::V - G::REQ - G::SP, (Q * D - 1) * G::REQ + 1>();
        next = pq_ring_read<Q>(C.ring, slot, C.lane);
    }
    const bool skip_alpha = !EXACT && (opaque & 0x3Fu) == 0x3Fu; // bit j: the row j before the newest is opaque
    float wv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        wv[j] = wnext[j];
        asm volatile("" : "+v"(wv[j])); // VGPR copy: scalar operands halve the VALU issue rate
    }
    {
        const uint32_t oyn = oy + 1 < A.oh ? oy + 1 : oy;
        cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane(oyn) * 6);
#pragma unroll
        for (int j = 0; j < 6; ++j) wnext[j] = wt[j];
    }
    pq_row<EXACT, P, Q, S + ADV, NARROW>(win, wv, W, C.rs, C.st, oy * row_bytes, skip_alpha);
    if constexpr (MOVES) {
        // the oldest row out, row r + 3 + ADV in; then request row r + 3 + ADV + Q D into the same ring slot
        const PxQ<Q> px = pq_swz<Q>(next, A.sel);
        if (!EXACT) opaque = (opaque << 1) | pq_row_is_opaque<Q>(px);
        pq_cvt_row<Q>(px, win[(S + ADV) % 6]);
        int rn = r + 3 + ADV + Q * D;
        rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
        // the slot is requested again only when its read has RETURNED (the converted row is an operand of this empty statement):
        // nothing orders a queued ds_read behind a later LDS-DMA write (see the x2 kernel)
        asm volatile("" : : "v"(win[(S + ADV) % 6][0]), "v"(win[(S + ADV) % 6][4 * Q - 1]) : "memory");
        pq_request_row<Q>(C.src, C.in_off + (uint32_t)rn * (A.iw * 4), C.ring_lds + slot);
    }
}

template <bool EXACT, int P, int Q, int S, int PH, bool NARROW>
__device__ __forceinline__ void pq_phases(float (&win)[6][4 * Q], const PqStepCtx &C, uint32_t pos, int r, const float (&W)[P][6],
                                          float (&wnext)[6], uint32_t &opaque)
{
    if constexpr (PH < P) {
        pq_phase<EXACT, P, Q, S, PH, NARROW>(win, C, pos, r, W, wnext, opaque);
        pq_phases<EXACT, P, Q, S, PH + 1, NARROW>(win, C, pos, r, W, wnext, opaque);
    }
}

// One group of input rows (r .. r + Q - 1), r a multiple of Q -> output rows P r / Q .. + P - 1.  At entry window slot (S + j) % 6
// holds input row r - 3 + j and the ring's Q slots at `pos` hold rows r + 3 .. r + 2 + Q (requested kPqDepth steps ago).
// Vector memory instructions of a step, in issue order and on every path: per phase its SP stores, then -- where the window moves
// -- the REQ pieces of a row request.
template <bool EXACT, int P, int Q, int S, bool NARROW>
__device__ __forceinline__ void pq_step(float (&win)[6][4 * Q], const PqStepCtx &C, uint32_t &pos, int r, const float (&W)[P][6],
                                        float (&wnext)[6], uint32_t &opaque)
{
    using G = PqGeom<P, Q>;
    pq_phases<EXACT, P, Q, S, 0, NARROW>(win, C, pos, r, W, wnext, opaque);
    pos = pos + Q * G::kSlotBytes == (uint32_t)(Q * kPqDepth) * G::kSlotBytes ? 0u : pos + Q * G::kSlotBytes;
}

template <bool EXACT, int P, int Q, int U, bool NARROW>
__device__ __forceinline__ bool pq_steps(float (&win)[6][4 * Q], const PqStepCtx &C, uint32_t &pos, int rbase, int r_end,
                                         const float (&W)[P][6], float (&wnext)[6], uint32_t &opaque)
{
    // UNROLL steps so the rotating window indices are compile-time constants; the block leaves the loop after its last row group,
    // so every path through the loop carries the vector memory instructions the hand-counted waits assume
    if constexpr (U < PqGeom<P, Q>::UNROLL) {
        pq_step<EXACT, P, Q, (U * Q) % 6, NARROW>(win, C, pos, rbase + U * Q, W, wnext, opaque);
        if (rbase + (U + 1) * Q >= r_end) return true;
        return pq_steps<EXACT, P, Q, U + 1, NARROW>(win, C, pos, rbase, r_end, W, wnext, opaque);
    }
    return false;
}

// One wave loads a strip of 64 Q input columns (Q per lane; the HL first and last lanes are the halo of the others) and walks `th`
// input rows with a 6-row f32 window.
template <bool EXACT, int P, int Q, bool NARROW = false>
__global__ __launch_bounds__(256) void k_lanczos3_pq(const LanczosPqArgs A)
{
    using G = PqGeom<P, Q>;
    const int lane = threadIdx.x & (kWave - 1);
    // each XCD gets a contiguous run of (frame, row block, strips), as in the x2 kernel
    const uint32_t vid = xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t frame = vid / gridDim.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((vid % gridDim.x) * 4 + (threadIdx.x >> 6));
    if (wave >= A.nstrips * A.nrowblocks) return;
    const uint32_t strip = wave % A.nstrips;
    const uint32_t rb = wave / A.nstrips;
    const int c = (int)(strip * G::kStripCols) + (lane - G::HS) * Q; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - Q ? (int)A.iw - Q : cl;
    const uint8_t *src = A.in + (size_t)frame * A.in_frame_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)frame * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
    const uint32_t in_off = (uint32_t)cl * 4u; // byte offset of the lane's 4 Q bytes inside an input row
    const int r0 = (int)(rb * A.th); // a multiple of Q
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    auto row_off = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return in_off + (uint32_t)rr * (A.iw * 4);
    };
    __shared__ __attribute__((aligned(16))) uint8_t lds_rows[4][Q * kPqDepth * G::kSlotBytes];
    __shared__ __attribute__((aligned(16))) uint32_t lds_stage[4][kWave * P];
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&lds_rows[w][0]);

    // the horizontal weights of this lane's P output columns, for the whole walk (a lane outside the image takes the last group's:
    // it stores nothing)
    float W[P][6];
    {
        const uint32_t g = (uint32_t)(cl / Q);
        const float *wl = A.wx6 + (size_t)g * (6 * P);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j) W[p][j] = wl[p * 6 + j];
    }
    float win[6][4 * Q];
    uint32_t opaque = 0;
    {
        // the six rows of the first window (ordinary loads, all in flight together), then the first requests
        PxQ<Q> first[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) first[j] = *reinterpret_cast<const PxQ<Q> *>(src + row_off(r0 - 3 + j));
#pragma unroll
        for (int j = 0; j < Q * kPqDepth; ++j)
            pq_request_row<Q>(src, row_off(r0 + 3 + j), ring_lds + (uint32_t)j * G::kSlotBytes);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const PxQ<Q> px = pq_swz<Q>(first[j], A.sel);
            if (!EXACT) opaque = (opaque << 1) | pq_row_is_opaque<Q>(px);
            pq_cvt_row<Q>(px, win[j]);
        }
        // the hand-counted waits of the loop assume that nothing older than its own instructions is outstanding
        pq_wait_vmcnt<0, 0>();
    }
    // vertical weights of the block's first output row (those of every later row are loaded one row ahead)
    float wnext[6];
    {
        typedef const __attribute__((address_space(4))) float *cfloat_p;
        cfloat_p wt = (cfloat_p)(uintptr_t)(A.wy6 + (size_t)__builtin_amdgcn_readfirstlane((uint32_t)P * (uint32_t)(r0 / Q)) * 6);
#pragma unroll
        for (int j = 0; j < 6; ++j) wnext[j] = wt[j];
    }
    uint32_t pos = 0;
    PqStore st;
    st.stage = &lds_stage[w][0];
    st.widx = (uint32_t)((lane - G::HS) & (kWave - 1)) * P;
    st.lane = lane;
    {
        // the strip's row segment: dwords strip NS P .. of the output row, as many as lie inside the image (a multiple of 4)
        const uint32_t first = strip * (uint32_t)(G::NS * P), row_dwords = A.iw / Q * P;
        const uint32_t valid = row_dwords - first < (uint32_t)(G::NS * P) ? row_dwords - first : (uint32_t)(G::NS * P);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t d = 4u * (uint32_t)(lane + kWave * k);
            st.off[k] = d + 4u <= valid ? (first + d) * 4u : 0x80000000u;
        }
    }
    const PqStepCtx C = {A, &lds_rows[w][0], ring_lds, lane, src, rs, in_off, st};
    for (int rbase = r0; rbase < r_end; rbase += Q * G::UNROLL)
        if (pq_steps<EXACT, P, Q, 0, NARROW>(win, C, pos, rbase, r_end, W, wnext, opaque)) break;
}

template <int P, int Q>
hipError_t launch_pq(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave, bool narrow)
{
    using G = PqGeom<P, Q>;
    LanczosPqArgs A;
    A.wx6 = T.lz_wx6;
    A.wy6 = T.lz_wy6;
    A.sel = L.in_sel;
    A.iw = L.iw;
    A.ih = L.ih;
    A.oh = L.oh;
    A.nstrips = cdiv(L.iw, (uint32_t)G::kStripCols);
    A.th = rows_per_wave ? (rows_per_wave + Q - 1) / Q * Q : 24;
    A.nrowblocks = cdiv(L.ih, A.th);
    A.in_frame_bytes = launch_in_frame_bytes(L);
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    const uint32_t nwaves = A.nstrips * A.nrowblocks;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
        if (exact && narrow)
            hipLaunchKernelGGL((k_lanczos3_pq<true, P, Q, true>), grid, block, 0, L.stream, A);
        else if (exact)
            hipLaunchKernelGGL((k_lanczos3_pq<true, P, Q, false>), grid, block, 0, L.stream, A);
        else if (narrow)
            hipLaunchKernelGGL((k_lanczos3_pq<false, P, Q, true>), grid, block, 0, L.stream, A);
        else
            hipLaunchKernelGGL((k_lanczos3_pq<false, P, Q, false>), grid, block, 0, L.stream, A);
    });
}

} // namespace

// The Q = 5 factors are one translation unit each (nus_k_lanczos_pq_65.hip, _75, _85, _95: their six unrolled row groups are 80 - 230 KB of
// code per instantiation and a minute of compile time); nus_k_lanczos_pq.hip holds the others and the dispatch.
hipError_t launch_lanczos_pq_65(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave, bool narrow);
hipError_t launch_lanczos_pq_75(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave, bool narrow);
hipError_t launch_lanczos_pq_85(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave, bool narrow);
hipError_t launch_lanczos_pq_95(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave, bool narrow);

} // namespace nus

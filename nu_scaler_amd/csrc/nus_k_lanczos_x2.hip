// nus_k_lanczos_x2.hip -- exact-x2 separable resize (Lanczos-3, Catmull-Rom, Triangle): the dominant
// kernel of the 1080p -> 4K path.  image-0.24.9 imageops::resize as called at
// Nu_scale/src/upscale/common.rs:243-251 (vertical pass into f32, then horizontal pass).

// Cache-policy bits of the output stores (buffer-store aux operand on gfx950: 1 sc0, 2 nt, 16 sc1).  nt: nothing this kernel
// writes is read again by it, and the 75 MB a unit writes otherwise push the input rows -- which three waves read -- out of
// the L2.  With every store instruction writing one contiguous KiB (round 2) nt is a gain: one-launch step 6.05 -> 5.81 ms
// (-4 %) on a box with a low copy ceiling, 5.79 -> 5.67 (-2 %) on a fast one, the plain kernel -2.5 % on both
// (profiles/r04_x2_cache_policy_sweep.txt; sc1 / sc0 variants and nt on the row REQUESTS, which loses 5-7 %, next to it).
// (Round 1 had measured nt stores 1.6-1.8x slower -- on 16-byte pieces at a 32-byte stride.)
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 2
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
    // UNIT kernels only (one launch per pipeline step, see k_lanczos3_x2): second 4K output (the up-scaled in-between frames),
    // the in-between frames themselves (may be null), frames of the launch and the order the waves walk them in
    uint8_t *out_mid;
    uint8_t *mid;
    uint32_t nframes, order;
    uint32_t rb0; // plain kernels: first row block of the launch (UpscaleLaunch::row0 / th); nrowblocks counts from there
};

// Input rows of the x2 kernels.  BLEND 0: the frame itself.  BLEND 1 / 2: the zero-flow in-between
// frame of a pair (A, B) is formed on the fly -- trunc((1-t) a + t b) per channel, exactly the u8
// pixel k_blend_zero_flow would have stored (interpolation/mod.rs:407-411) -- so "interpolate, then
// upscale the interpolated frame" (nu_scaler_py/nu_scaler/main.py:999-1008) needs no round trip
// through HBM.  At t = 0.5 both products and the sum are exact and the truncation is a floor of a
// half-integer: one v_lerp_u8 per pixel.

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int BLEND>
struct RowRaw {
    u32x4 a, b;
};
template <>
struct RowRaw<0> {
    u32x4 a;
};

// Row prefetches of the main loop.  gfx950 retires vector memory operations in issue order on ONE
// counter (vmcnt) -- loads and stores together.  With ordinary loads the compiler, merging the three code
// paths of a step (border rows / opaque rows / rows with alpha) around the loop's back edge, settled for
// `s_waitcnt vmcnt(2)` / `vmcnt(3)` in front of every row conversion: each wave then drained all but its
// two or three newest STORES once per step, one phase after issuing them, and the kernel's SIMD time and
// its store time added up instead of overlapping (round 1: 0.47 of the HBM roofline).
// So the prefetches are taken out of the compiler's sight and counted by hand:
//  * a row is requested with an LDS-DMA load (global_load_lds_dwordx4: 16 B per lane straight into a
//    per-wave 1-KiB LDS slot, no VGPR destination) issued from inline assembly -- the compiler's s_waitcnt
//    insertion does not see it, and there is no destination register it could copy before the data is in;
//  * before the slot is read (an ordinary ds_read_b128) the wave waits with a hand-counted
//    `s_waitcnt vmcnt(N)`, N = the vector memory instructions issued since the row's request, which is the
//    same on every path through a step (lanczos_x2_step); nothing else waits on vmcnt inside the loop.
// tools/check_hidden_loads.py verifies both properties on the generated code (tests/test_kernel_asm.py).
#ifndef NUS_LZ_ASM_LOADS
#define NUS_LZ_ASM_LOADS 1 // dev macro: 0 = plain loads into registers, compiler-placed waits (A/B timing only)
#endif
#ifndef NUS_LZ_DEPTH
#define NUS_LZ_DEPTH 2 // prefetch distance in steps (1, 2, 3 or 6: must divide the 6-way unrolled loop)
#endif
#ifndef NUS_LZ_WAIT_EARLY
#define NUS_LZ_WAIT_EARLY 1 // 1: wait + LDS read between an even row's two passes (the read's latency hides behind the
                            // horizontal pass); 0: after the even row's stores
#endif
#ifndef NUS_LZ_ABLATE
#define NUS_LZ_ABLATE 0 // dev macro, timing only (wrong pixels): 1 no stores, 2 no arithmetic, 3 stores only (no loads either),
                        // 5 every row request goes to the first 16 rows of frame 0 (cache hits), 6 no lane exchange, 7 no pack
#endif
#if (NUS_LZ_ABLATE != 0 || !NUS_LZ_ASM_LOADS) && !defined(NUS_DEV_BUILD)
#error "timing-only dev macros of k_lanczos3_x2 (wrong pixels / A-B forms) need -DNUS_DEV_BUILD: never in a product build"
#endif
constexpr int kLzDepth = NUS_LZ_DEPTH;
static_assert(6 % kLzDepth == 0, "prefetch distance must divide 6");

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 is a reserved register: nothing else in these kernels uses it
// Request 16 B per lane from `base + off` (base wave-uniform, off < 4 GiB) into the 1-KiB LDS slot at
// byte offset `lds` (wave-uniform): lane l lands at lds + 16 l.
#ifndef NUS_LZ_LOAD_POLICY
#define NUS_LZ_LOAD_POLICY 0 // cache policy of the row requests: 0 default, 1 nt, 2 sc1, 3 sc0 sc1 (tuning knob)
#endif
#if NUS_LZ_LOAD_POLICY == 1
#define NUS_LZ_LOAD_MOD " nt"
#elif NUS_LZ_LOAD_POLICY == 2
#define NUS_LZ_LOAD_MOD " sc1"
#elif NUS_LZ_LOAD_POLICY == 3
#define NUS_LZ_LOAD_MOD " sc0 sc1"
#else
#define NUS_LZ_LOAD_MOD ""
#endif
__device__ __forceinline__ void dma_row16(const uint8_t *base, uint32_t off, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" NUS_LZ_LOAD_MOD : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// Wait until at most N of this wave's vector memory instructions are outstanding: in-order retirement, so
// everything older than its N youngest has landed.  "memory": no load, store or LDS access moves across it.
// BACK (for tools/check_hidden_loads.py): the wait is for the BACK-th most recent row request; 0 = all.
template <int N, int BACK>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; nus-wait back=%1" : : "n"(N), "n"(BACK) : "memory");
}

// LDS ring of one wave: kLzDepth slots of NL rows (NL = 2 when a pair is blended on load), 64 lanes x 16 B each.
template <int BLEND>
struct RowRing {
    static constexpr int NL = BLEND ? 2 : 1;
    u32x4 (*slot)[64]; // [kLzDepth * NL][64], this wave's part of the block's LDS
    uint32_t lds;      // its byte offset in LDS (wave-uniform)
    int lane;

    __device__ __forceinline__ void request(int k, const uint8_t *pa, const uint8_t *pb, uint32_t off) const
    {
        dma_row16(pa, off, lds + (uint32_t)(k * NL) * 1024u);
        if constexpr (BLEND != 0) dma_row16(pb, off, lds + (uint32_t)(k * NL + 1) * 1024u);
    }
    __device__ __forceinline__ RowRaw<BLEND> read(int k) const
    {
        RowRaw<BLEND> r;
        r.a = slot[k * NL][lane];
        if constexpr (BLEND != 0) r.b = slot[k * NL + 1][lane];
        return r;
    }
};

// ordinary loads into registers (compiler-placed waits): the first window of a block, the edge kernel
template <int BLEND>
__device__ __forceinline__ RowRaw<BLEND> fetch_row_plain(const uint8_t *pa, const uint8_t *pb, size_t off)
{
    RowRaw<BLEND> r;
    r.a = *reinterpret_cast<const u32x4 *>(pa + off);
    if constexpr (BLEND != 0) r.b = *reinterpret_cast<const u32x4 *>(pb + off);
    return r;
}

__device__ __forceinline__ uint4 as_uint4(const u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }

// (the blend is per channel, so the channel swizzle is applied once, to the blended pixel)
template <int BLEND>
__device__ __forceinline__ uint4 resolve_row(const RowRaw<BLEND> &r, float t, uint32_t sel)
{
    if constexpr (BLEND == 0) {
        return swz4(as_uint4(r.a), sel);
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
#if NUS_LZ_ABLATE == 6
    return v;
#endif
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_down(float v) // value of lane+1
{
#if NUS_LZ_ABLATE == 6
    return v;
#endif
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

// Rows whose alpha is ONE value over everything the wave holds of them (all 64 lanes x 4 columns) -- and the same value over
// the whole 6-row tap window -- take the 3-channel path below: the taps are normalised (sum(w) is 1 within a few f32 ulps on
// both axes), so the alpha the CPU computes for such a window is round(A (1 + eps)), |A eps| < 1e-3: A itself, which is stored as
// a constant and a quarter of the per-pixel arithmetic is skipped.  Captured and rendered frames are opaque (A = 255, rounds 1-4's
// "opaque path"); overlays and borders with a flat alpha (0, 128, ..) qualify as well since round 5; frames whose alpha varies
// inside a window take the 4-channel path for that window.
// AlphaRun: eq bit j = "input row (newest - j) is flat and so is the row before it, with the same alpha" -- five set bits are six
// rows of one alpha; `alpha` = the newest row's alpha in bits 31:24 (meaningful while eq bit 0 is set).  All wave-uniform (SGPRs).
#ifndef NUS_OPAQUE_PATH
#define NUS_OPAQUE_PATH 1 // dev macro: 0 builds the x2 kernel without the 3-channel path (A/B timing only)
#endif
#ifndef NUS_BLEND_OPAQUE_PATH
#define NUS_BLEND_OPAQUE_PATH 1 // dev macro: 0 = the blend variants without the 3-channel path, as in round 2 (A/B timing only)
#endif
#ifndef NUS_FLAT_ALPHA
#define NUS_FLAT_ALPHA 1 // dev macro: 0 = rounds 1-4: only alpha == 255 takes the 3-channel path (A/B timing only)
#endif
#if (!NUS_OPAQUE_PATH || !NUS_BLEND_OPAQUE_PATH || !NUS_FLAT_ALPHA) && !defined(NUS_DEV_BUILD)
#error "timing-only dev macros of k_lanczos3_x2 need -DNUS_DEV_BUILD: never in a product build"
#endif
#define NUS_LZ_BLEND_OPAQUE_PATH_OK(BLEND) ((BLEND) == 0 || NUS_BLEND_OPAQUE_PATH != 0)
struct AlphaRun {
    uint32_t eq = 0;    // history of "flat, and equal to the row before"
    uint32_t alpha = 0; // alpha of the newest row << 24 (if it was flat)
    uint32_t flat = 0;  // the newest row was flat
};
__device__ __forceinline__ void alpha_run_push(AlphaRun &R, const uint4 raw)
{
#if !NUS_OPAQUE_PATH
    return;
#endif
#if NUS_FLAT_ALPHA
    // the lane's four alphas agree (the xors leave nothing in the top byte) and equal the first lane's
    const uint32_t first = __builtin_amdgcn_readfirstlane(raw.x) & 0xFF000000u;
    const bool lane_ok = (((raw.x ^ raw.y) | (raw.x ^ raw.z) | (raw.x ^ raw.w) | (raw.x ^ first)) & 0xFF000000u) == 0u;
#else
    const uint32_t first = 0xFF000000u;
    const bool lane_ok = (raw.x & raw.y & raw.z & raw.w) >= 0xFF000000u;
#endif
    const uint32_t flat = __builtin_amdgcn_ballot_w64(!lane_ok) == 0ull ? 1u : 0u;
    R.eq = (R.eq << 1) | (flat & R.flat & (first == R.alpha ? 1u : 0u));
    R.alpha = first;
    R.flat = flat;
}
// the six newest rows carry one alpha: the 3-channel path may be taken, storing R.alpha
__device__ __forceinline__ bool alpha_run_six(const AlphaRun &R) { return (R.eq & 0x1Fu) == 0x1Fu; }

// Vertical pass of one output row: 6 taps from window slots BASE .. BASE+5 (mod 6).
// NARROW: the filter's outermost taps (slots 0 and 5 of both phases) carry weight 0 on both axes (Catmull-Rom at x2: four taps;
// host-checked) and are skipped -- adding v * 0 changes no sum, so the bits are those of the six-tap form.
template <bool EXACT, int BASE, bool ALPHA, bool NARROW = false>
__device__ __forceinline__ void lanczos_x2_vpass(const float (&win)[6][16], const float (&w)[6], float (&V)[16])
{
#if NUS_LZ_ABLATE == 2 || NUS_LZ_ABLATE == 3
#pragma unroll
    for (int k = 0; k < 16; ++k) V[k] = win[BASE % 6][k];
    return;
#endif
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (!ALPHA && (k & 3) == 3) continue; // V[alpha] is not read by the 3-channel horizontal pass
        constexpr int J0 = NARROW ? 1 : 0, J1 = NARROW ? 5 : 6;
        float acc = win[(BASE + J0) % 6][k] * w[J0]; // == fma(.., 0) and a VOP2 instruction
#pragma unroll
        for (int j = J0 + 1; j < J1; ++j) acc = mac_tight<EXACT>(acc, win[(BASE + j) % 6][k], w[j]);
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
        for (int j = 1; j < 6; ++j) acc = mac_tight<EXACT>(acc, win[(BASE + j) % 6][k], w[j]);
        V[k] = acc;
    }
}

// Horizontal pass of the lane's 8 output pixels, convert + pack.
template <bool EXACT, bool ALPHA, bool NARROW = false>
__device__ __forceinline__ void lanczos_x2_hpass(const float (&V)[16], const PhaseWeights &W, uint32_t (&o)[8], uint32_t flat_alpha = 0u)
{
    const uint32_t a0 = ALPHA ? 0u : flat_alpha; // 3-channel path: the window's one alpha (wave-uniform), already in bits 31:24
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = a0;
#if NUS_LZ_ABLATE == 2 || NUS_LZ_ABLATE == 3
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = __float_as_uint(V[q]);
    return;
#endif
#pragma unroll
    for (int c = 0; c < (ALPHA ? 4 : 3); ++c) {
        float e[10]; // vertical sums of input columns c0-3 .. c0+6 for this channel (NARROW: e[0] and e[9] are not read)
        e[0] = NARROW ? 0.0f : lane_up(V[1 * 4 + c]);
        e[1] = lane_up(V[2 * 4 + c]);
        e[2] = lane_up(V[3 * 4 + c]);
        e[3] = V[0 * 4 + c];
        e[4] = V[1 * 4 + c];
        e[5] = V[2 * 4 + c];
        e[6] = V[3 * 4 + c];
        e[7] = lane_down(V[0 * 4 + c]);
        e[8] = lane_down(V[1 * 4 + c]);
        e[9] = NARROW ? 0.0f : lane_down(V[2 * 4 + c]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            constexpr int J0 = NARROW ? 1 : 0, J1 = NARROW ? 5 : 6;
            float ae = e[m + J0] * W.e[J0];
            float ao = e[m + 1 + J0] * W.o[J0];
#pragma unroll
            for (int j = J0 + 1; j < J1; ++j) {
                ae = mac_tight<EXACT>(ae, e[m + j], W.e[j]);
                ao = mac_tight<EXACT>(ao, e[m + 1 + j], W.o[j]);
            }
#if NUS_LZ_ABLATE == 7
            o[2 * m] += __float_as_uint(ae);
            o[2 * m + 1] += __float_as_uint(ao);
#else
            o[2 * m] = pack_u8<EXACT>(ae, c, o[2 * m]);
            o[2 * m + 1] = pack_u8<EXACT>(ao, c, o[2 * m + 1]);
#endif
        }
    }
}

// Where a wave's output row goes.  A lane computes 8 consecutive output pixels (32 B), but a store instruction in
// which every lane writes 16 B at a 32-B stride leaves each 128-B line half written until the second
// instruction fills the other half -- and on gfx950 such half-line writes, once HBM reads are in the mix,
// cost almost twice the time of whole-line writes (tools/probe_rw_mix.hip: 10.4 vs 7.6 us per 1080p -> 4K
// frame for the same bytes).  So the row's 2 KiB are turned round in LDS first -- each lane writes its 32 B at
// 32 * lane, then reads 16 B at 16 * lane and at 1024 + 16 * lane -- and each store instruction writes one
// contiguous KiB.  LDS instructions of one wave execute in order, so no barrier is involved.
struct RowStore {
    u32x4 *stage;          // this wave's 2 KiB of LDS (128 x 16 B)
    uint32_t off_a, off_b; // byte offset of this lane's 16 B inside an output row for the two stores; pieces that
                           // belong to halo lanes or to the edge kernel's columns sit at 2^31: the range check drops them
    int idx_a, idx_b;      // which 16-B pieces of the stage this lane reads back
    int lane;
};

#ifndef NUS_LZ_CONTIG_STORES
#define NUS_LZ_CONTIG_STORES 1 // dev macro: 0 = each lane stores its own 2 x 16 B (A/B timing only)
#endif
// Round 6: the same contiguous KiB per store instruction WITHOUT the turn through LDS.  gfx950's v_permlane32_swap_b32 vdst, src
// swaps lanes 32..63 of vdst with lanes 0..31 of src.  With vdst = the lane's first 16 bytes (A) and src = its second 16 (B), four
// swaps (one per dword) leave in A the 32 bytes of each of lanes 0..31 -- A of lane l in lane l, B of lane l in lane l + 32 -- and
// in B those of lanes 32..63: each store instruction still writes ONE contiguous KiB, in an interleaved lane order (lane l < 32 at
// 32 l, lane l >= 32 at 32 (l - 32) + 16), which the memory system takes at the rate of the lane-ordered KiB
// (tools/probe_row_store_shapes.hip: 7.1 / 7.1 us per frame of stores + loads against 9.7 for the lane's own strided pieces).
// No LDS write, no LDS read, no lgkmcnt wait between a row's arithmetic and its stores; 8 KiB of LDS per block less.  Built, bit-identical
// (385 GPU parity cases), and measured against the LDS turn in one process: one-launch step 5.691 against 5.693 ms per 300 units
// (gradient), 6.751 / 6.787 (noise); the plain kernel 2.667 / 2.647 and 3.310 / 3.310 ms per 300 frames
// (profiles/r06_x2_row_store_without_lds_ab.txt).  The turn through LDS was not what held the kernel: it sits on its memory and
// issue floors either way.  The product keeps the LDS turn, whose stores cover whole 128-byte lines per
// instruction (with 60 storing lanes the swap form splits one line in fifteen between its two stores).
#ifndef NUS_LZ_SWAP_STORES
#define NUS_LZ_SWAP_STORES 0 // 1 = the permlane32_swap form (A/B builds)
#endif

// The two 16-B stores of an output row.  Buffer stores: lanes that must not write carry an offset beyond
// num_records and the hardware range check drops them, so the store instructions issue on every path and
// for every lane -- the hand-counted waits of the row prefetches (wait_vmcnt) rely on exactly two per phase.
__device__ __forceinline__ void lanczos_x2_store(const uint32_t (&o)[8], __amdgpu_buffer_rsrc_t rs, const RowStore &st,
                                                 uint32_t row_off)
{
#if NUS_LZ_ABLATE == 1
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : : "v"(o[q]));
    return;
#endif
    u32x4 lo = {o[0], o[1], o[2], o[3]}, hi = {o[4], o[5], o[6], o[7]};
#if NUS_LZ_CONTIG_STORES && NUS_LZ_SWAP_STORES
    typedef uint32_t u32x2_sw __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x2_sw r = __builtin_amdgcn_permlane32_swap(lo[k], hi[k], false, false); // (the compiler pads the VALU -> swap hazard)
        lo[k] = r.x, hi[k] = r.y;
    }
#elif NUS_LZ_CONTIG_STORES
    st.stage[2 * st.lane] = lo;
    st.stage[2 * st.lane + 1] = hi;
    __builtin_amdgcn_wave_barrier(); // compiler only: the reads below see other lanes' writes (same wave, in-order LDS)
    lo = st.stage[st.idx_a];
    hi = st.stage[st.idx_b];
    __builtin_amdgcn_wave_barrier();
#endif
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, row_off + st.off_a, 0, NUS_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, row_off + st.off_b, 0, NUS_STORE_AUX);
}

// One input row r -> output rows 2r (taps r-3..r+2) and 2r+1 (taps r-2..r+3).
// At entry window slot (S+j)%6 holds input row r-3+j, j = 0..5, and ring slot S % D holds row r+3 (requested D
// steps ago).  Row r-3 dies after the even phase, so row r+3 is converted into its slot BETWEEN the two
// phases: only 6 rows (96 VGPRs) are ever live, not 7.
//
// Vector memory instructions of a step, in issue order: 2 stores (even row), NL row requests (row r+3+D),
// 2 stores (odd row) -- on every code path: the stores are range-checked buffer stores that always issue, and
// they sit in the straight-line code after the paths have joined.  Issued since row r+3's requests when the
// wave waits for it: the odd stores of step S-D (2), D-1 whole steps (4 + NL each) and, if the wait comes
// after them, this step's even stores (2): N = 4 D + (D - 1) NL - (early ? 2 : 0).  Besides the row the wait
// only covers stores at least D steps old.
//
// UNIT kernels (the in-between frames are an output too): one more store per step, the row just resolved (row r+3, while it
// belongs to this wave's row block: MidStore), between the even stores and the requests -- M = 1 more instruction per
// whole step in the count: N = 4 D + (D - 1)(NL + M) - (early ? 2 : 0).
struct MidStore {
    __amdgpu_buffer_rsrc_t rs; // the in-between frame (num_records 0 in the real-frame role and without a buffer: every store dropped)
    uint32_t lane_off;         // this lane's 16 B inside a row; 2^31 for halo lanes and lanes beyond the image
    int r_end;                 // rows below r_end are the wave's own
};

template <bool UNIT>
__device__ __forceinline__ void lanczos_x2_store_mid(const uint4 px, const MidStore &ms, int row, uint32_t row_bytes_in)
{
    if constexpr (UNIT) {
        // not the wave's row: an offset that stays out of range (and below 2^32) with either kind of lane_off added
        const uint32_t row_off = row < ms.r_end ? (uint32_t)row * row_bytes_in : 0x7FFFFFF0u;
        u32x4 v = {px.x, px.y, px.z, px.w};
        __builtin_amdgcn_raw_buffer_store_b128(v, ms.rs, row_off + ms.lane_off, 0, NUS_STORE_AUX);
    }
}

template <bool EXACT, int BLEND, bool UNIT, int S, bool NARROW = false>
__device__ __forceinline__ void lanczos_x2_step(float (&win)[6][16], const RowRing<BLEND> &ring, RowRaw<BLEND> (&raw)[2],
                                                AlphaRun &arun, int r, uint32_t in_off, const RowStore &st,
                                                const LanczosX2Args &A, const PhaseWeights &W, const uint8_t *src,
                                                const uint8_t *src_b, __amdgpu_buffer_rsrc_t rs, float t, const MidStore &ms)
{
    constexpr int D = kLzDepth, NL = BLEND ? 2 : 1, M = UNIT ? 1 : 0;
    constexpr bool HIDDEN = NUS_LZ_ASM_LOADS != 0, EARLY = HIDDEN && NUS_LZ_WAIT_EARLY != 0;
    const uint32_t row_bytes = A.iw * 8; // output row: 2*iw pixels
    const uint32_t off0 = (uint32_t)(2 * r) * row_bytes;
    const bool interior = r >= 4 && r + 5 <= (int)A.ih; // wave-uniform
    // The 3-channel path is compiled into the FMA-mode kernels (plain and blend variants: 164 - 166 VGPRs, three waves per SIMD
    // since the LDS-DMA ring took the in-flight rows out of the registers); EXACT is the register-hungry verification mode.
    constexpr bool OP = !EXACT && NUS_LZ_BLEND_OPAQUE_PATH_OK(BLEND);
    float V[16];
    uint32_t o[8];
    RowRaw<BLEND> next; // row r+3
    // the six newest rows are this phase's taps
    const uint32_t path = !interior ? 0u : ((OP && alpha_run_six(arun)) ? 1u : 2u); // wave-uniform
    const uint32_t alpha_e = arun.alpha;
    if (path == 0)
        lanczos_x2_vpass_edge<EXACT, S>(win, A.wy6, 2 * (uint32_t)r, V);
    else if (path == 1)
        lanczos_x2_vpass<EXACT, S, false, NARROW>(win, W.e, V);
    else
        lanczos_x2_vpass<EXACT, S, true, NARROW>(win, W.e, V);
#if NUS_LZ_ABLATE == 1
    if (EARLY) {
        wait_vmcnt<(D - 1) * (NL + M), (D - 1) * NL + 1>();
        next = ring.read(S % D);
    }
#elif NUS_LZ_ABLATE == 3
    next = RowRaw<BLEND>{};
#else
    if (EARLY) {
        wait_vmcnt<4 * D + (D - 1) * (NL + M) - 2, (D - 1) * NL + 1>();
        next = ring.read(S % D);
    }
#endif
    if (path == 1)
        lanczos_x2_hpass<EXACT, false, NARROW>(V, W, o, alpha_e);
    else
        lanczos_x2_hpass<EXACT, true, NARROW>(V, W, o);
    lanczos_x2_store(o, rs, st, off0); // after the paths have joined: straight-line code holds every memory instruction
    // row r+3 in, then request row r+3+D into the same slot
    if (HIDDEN && !EARLY) {
        wait_vmcnt<4 * D + (D - 1) * (NL + M), (D - 1) * NL + 1>();
        next = ring.read(S % D);
    }
    if (!HIDDEN) next = raw[S & 1];
    {
        const uint4 px = resolve_row<BLEND>(next, t, A.sel);
        if (OP) alpha_run_push(arun, px);
        cvt_row(px, win[S % 6]);
        lanczos_x2_store_mid<UNIT>(px, ms, r + 3, A.iw * 4);
    }
    {
        int rn = r + 3 + (HIDDEN ? D : 2);
        rn = rn < (int)A.ih - 1 ? rn : (int)A.ih - 1;
#if NUS_LZ_ABLATE == 5
        const uint32_t off = in_off + (uint32_t)(rn & 15) * (A.iw * 4);
        if (HIDDEN) ring.request(S % D, A.in, A.in, off);
        else
#elif NUS_LZ_ABLATE == 3
        const uint32_t off = 0;
        if (HIDDEN) {}
        else
#else
        const uint32_t off = in_off + (uint32_t)rn * (A.iw * 4);
        if (HIDDEN) {
            // the slot is requested again only when its reads have RETURNED (the converted row is an operand of this empty
            // statement, and volatile statements keep their order): nothing orders a queued ds_read behind a later LDS-DMA write
            // (seen in k_resize_down, whose waves keep the LDS busy: profiles/r03_resize_down_lds_dma_ring_ab.txt)
            asm volatile("" : : "v"(win[S % 6][0]), "v"(win[S % 6][15]) : "memory");
            ring.request(S % D, src, src_b, off);
        } else
#endif
            raw[S & 1] = fetch_row_plain<BLEND>(src, src_b, off);
    }
    const uint32_t path_o = !interior ? 0u : ((OP && alpha_run_six(arun)) ? 1u : 2u);
    if (path_o == 0) {
        lanczos_x2_vpass_edge<EXACT, S + 1>(win, A.wy6, 2 * (uint32_t)r + 1, V);
        lanczos_x2_hpass<EXACT, true, NARROW>(V, W, o);
    } else if (path_o == 1) {
        lanczos_x2_vpass<EXACT, S + 1, false, NARROW>(win, W.o, V);
        lanczos_x2_hpass<EXACT, false, NARROW>(V, W, o, arun.alpha);
    } else {
        lanczos_x2_vpass<EXACT, S + 1, true, NARROW>(win, W.o, V);
        lanczos_x2_hpass<EXACT, true, NARROW>(V, W, o);
    }
    lanczos_x2_store(o, rs, st, off0 + row_bytes);
}

// Exact x2 Lanczos-3.  One wave loads a strip of 256 input columns (4 per lane); lanes 1 .. 60 produce the
// strip's 240 input = 480 output columns, lanes 0 and 61 are their halo, lanes 62 / 63 idle (a strip of 240
// makes every output row of a strip 15 whole 128-B lines).  The wave walks `th` input rows, keeping a 6-row
// f32 window of its columns in registers:
//   row prefetch   : LDS-DMA into a per-wave ring, D steps ahead, hand-counted vmcnt waits (see RowRing)
//   vertical pass  : 6 taps from the register window
//   horizontal pass: 6 taps over the lane's own 4 columns + 3 columns from each
//                    neighbouring lane, fetched with wave_shr/wave_shl DPP moves
//   store          : the row's 2 KiB are turned round in LDS, two stores of one contiguous KiB each (RowStore)
// so every input byte is read once per strip-row-block and no workgroup barrier is needed.  (The 6 halo rows two vertically
// neighbouring row blocks share are requested a whole block apart in time; letting odd row blocks walk bottom-up through a
// vertical mirror of the frame, so that both readers of a halo reach it together, was built and measured in round 3: outputs
// identical, 9.31 against 9.41 us per frame, nothing on the unit step -- profiles/r03_lanczos_mirrored_row_blocks_ab.txt; not kept.)
// The 8 left-most and right-most output columns (renormalised edge weights) belong to
// k_lanczos3_x2_edges, which is launched behind this kernel and overwrites what it wrote there.
//
// UNIT = true (BLEND 1 / 2 only): ONE launch does a whole pipeline step over a batch of pairs (A_k, B_k) -- the up-scaled real
// frame, the up-scaled in-between frame and the in-between frame itself -- with the very same instruction stream.  Every
// (frame, row block, strip) is walked by two waves, and what tells them apart sits in SGPRs only:
//   role 0 (real frame)      : both row streams come from A_k (the blend of a row with itself is the row: v_lerp_u8(a, a) = a,
//                              and t = 0 at BLEND 2), output to A.out, in-between stores dropped (num_records 0).  (Requesting
//                              the row once and keeping the second request's place in the vmcnt count with a dropped store was
//                              measured: -0.5 %, profiles/r03_unit_kernel_one_dma_ab.txt; not worth a second instruction stream);
//   role 1 (in-between frame): rows of A_k and B_k blended on load as in the BLEND kernels, output to A.out_mid, and each
//                              resolved row of the wave's own block stored to A.mid.
// The two waves of a (frame, row block, strip) run side by side (consecutive workgroups of one XCD), so A_k's rows reach the
// second of them from cache, and with order = 1 (row-block-major) so do the rows of B_k = A_(k+1): the step reads every input
// row from HBM about once where three launches (blend, upscale, upscale) read it four times, and writes nothing twice.
// (NARROW: the compiler gives the four-tap form 215 registers, two waves per SIMD; held to three (168 registers) it spills 49
// dwords, whose reloads bring vmcnt waits back into the loop: 9.7 against 7.7 - 8.0 us per frame.  Left as the compiler has it.
// The six-tap instantiations are instruction for instruction what they were before the parameter existed.)
template <bool EXACT, int BLEND, bool UNIT, bool NARROW = false>
__global__ __launch_bounds__(256) void k_lanczos3_x2(const LanczosX2Args A)
{
    static_assert(!UNIT || BLEND != 0, "the unit kernel is a blend kernel whose real-frame role blends a row with itself");
    const int lane = threadIdx.x & (kWave - 1);
    // each XCD gets a contiguous run of (frame, row block, strips): neighbouring waves' halo rows and columns in its L2
    const uint32_t vid = xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    uint32_t frame, strip, rb, role = 0;
    if constexpr (UNIT) {
        // 1-D grid over nframes x nrowblocks x 2 roles x nstrips waves
        const uint32_t gw = __builtin_amdgcn_readfirstlane(vid * 4 + (threadIdx.x >> 6));
        const uint32_t per_rb = 2 * A.nstrips; // waves of one (frame, row block): the strips of role 0, then those of role 1
        if (gw >= per_rb * A.nrowblocks * A.nframes) return;
        const uint32_t grp = gw / per_rb, in_grp = gw % per_rb;
        if (A.order == 0) { // frame-major: row blocks of a frame in a run (vertical neighbours share their halo rows in L2)
            frame = grp / A.nrowblocks;
            rb = grp % A.nrowblocks;
        } else { // row-block-major: the same row block of consecutive frames in a run (B_k = A_(k+1) shared as well)
            rb = grp / A.nframes;
            frame = grp % A.nframes;
        }
        role = in_grp / A.nstrips;
        strip = in_grp % A.nstrips;
    } else {
        frame = vid / gridDim.x;
        const uint32_t wave = __builtin_amdgcn_readfirstlane((vid % gridDim.x) * 4 + (threadIdx.x >> 6));
        if (wave >= A.nstrips * A.nrowblocks) return;
        strip = wave % A.nstrips;
        rb = wave / A.nstrips + A.rb0;
    }
    const int c = (int)(strip * kLanczosX2StripCols) - 4 + lane * 4; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - 4 ? (int)A.iw - 4 : cl;
    const uint8_t *src = A.in + (size_t)frame * A.in_frame_bytes;
    const uint8_t *src_b = BLEND ? A.in_b + (size_t)frame * A.in_b_frame_bytes : src;
    float t = A.t;
    uint8_t *out = A.out;
    MidStore ms;
    if constexpr (UNIT) {
        const bool real = role == 0; // wave-uniform
        src_b = real ? src : src_b;
        t = real ? 0.0f : t; // BLEND 2: 1 * a + 0 * a = a exactly
        out = real ? A.out : A.out_mid;
        const bool owner = lane >= 1 && lane <= (int)(kLanczosX2StripCols / 4) && c >= 0 && c + 4 <= (int)A.iw;
        ms.lane_off = owner ? (uint32_t)c * 4u : 0x80000000u;
        // the in-between frames are tightly packed whatever the stride of the source frames (nuscaler_hip.h: d_mid)
        ms.rs = __builtin_amdgcn_make_buffer_rsrc(A.mid + (size_t)frame * ((size_t)A.iw * A.ih * 4u), 0,
                                                  real || A.mid == nullptr ? 0u : A.iw * A.ih * 4u, 0x00020000);
    } else {
        ms.lane_off = 0;
        ms.rs = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
    }
    // one buffer resource per output frame (< 2 GiB, checked by the host)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)frame * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
    // lane L computes the 8 output pixels of input columns c .. c+3; it stores them unless it is a halo lane or its
    // columns are the edge kernel's
#ifndef NUS_LZ_ALIGNED_STORES
#define NUS_LZ_ALIGNED_STORES 1 // the two stores start at the first STORING lane's pixels, not at the halo lane's: with 240-column
                                // strips every store instruction covers whole 128-B lines (0 = dev macro, A/B: -3 % on the opaque
                                // stream together with the edge-column overwrite, profiles/r02_lanczos_x2_variants_ab.txt)
#endif
#ifndef NUS_LZ_MAIN_WRITES_EDGES
#define NUS_LZ_MAIN_WRITES_EDGES 0 // rounds 2-3 (1): the main kernel also wrote the 8 edge columns per side (wrong: interior
                                   // weights) and the edge pass, launched BEHIND it, overwrote them.  Round 4 (0): it leaves them
                                   // alone (the two 16-byte pieces per row end are range-checked away), so the edge pass depends
                                   // on nothing the main kernel does and runs BESIDE it on a second stream (HipUpscaler::enqueue).
#endif
    auto computes_stored_pixels = [&](int L) {
        const int cc = (int)(strip * kLanczosX2StripCols) - 4 + L * 4;
#if NUS_LZ_ALIGNED_STORES && NUS_LZ_MAIN_WRITES_EDGES
        return L >= 1 && L <= (int)(kLanczosX2StripCols / 4) && cc >= 0 && cc + 4 <= (int)A.iw;
#else
        return L >= 1 && L <= (int)(kLanczosX2StripCols / 4) && cc >= 4 && cc + 8 <= (int)A.iw;
#endif
    };
    RowStore st;
    st.lane = lane;
#if NUS_LZ_CONTIG_STORES && NUS_LZ_SWAP_STORES
    {
        // after the swaps: store A carries lane l's first piece (l < 32) or lane l - 32's second piece; store B lane l + 32's first
        // (l < 32) or lane l's second -- each piece dropped (offset 2^31) when the lane that COMPUTED it stores nothing
        const int span0 = ((int)(strip * kLanczosX2StripCols) - 4) * 8; // byte offset of the span in an output row (may be < 0)
        const int la = lane < 32 ? lane : lane - 32, lb = lane < 32 ? lane + 32 : lane;
        const int half = lane < 32 ? 0 : 16;
        st.off_a = computes_stored_pixels(la) ? (uint32_t)(span0 + 32 * la + half) : 0x80000000u;
        st.off_b = computes_stored_pixels(lb) ? (uint32_t)(span0 + 32 * lb + half) : 0x80000000u;
        st.idx_a = st.idx_b = 0;
    }
#elif NUS_LZ_CONTIG_STORES
    {
        // after the turn in LDS this lane holds 16 B of the wave's 512-pixel span for each store: SKIP bytes into the span
        // for the first store, 1024 further for the second; they were computed by lanes (SKIP + 16 lane) / 32 and
        // (SKIP + 1024 + 16 lane) / 32
        constexpr int SKIP = NUS_LZ_ALIGNED_STORES ? 32 : 0;
        const int span0 = ((int)(strip * kLanczosX2StripCols) - 4) * 8; // byte offset of the span in an output row (may be < 0)
        const int pa = SKIP + 16 * lane, pb = SKIP + 1024 + 16 * lane;
        st.idx_a = pa / 16;
        st.idx_b = pb / 16 < 128 ? pb / 16 : 127; // (pieces past the span belong to no lane and are dropped)
        st.off_a = computes_stored_pixels(pa / 32) ? (uint32_t)(span0 + pa) : 0x80000000u;
        st.off_b = pb / 32 < 64 && computes_stored_pixels(pb / 32) ? (uint32_t)(span0 + pb) : 0x80000000u;
    }
#else
    st.off_a = computes_stored_pixels(lane) ? (uint32_t)c * 8u : 0x80000000u;
    st.off_b = st.off_a + 16u;
    st.idx_a = st.idx_b = 0;
#endif
    const uint32_t in_off = (uint32_t)cl * 4u; // the lane's byte offset inside an input row
    const int r0 = (int)(rb * A.th);
    const int r_end = (r0 + (int)A.th) < (int)A.ih ? (r0 + (int)A.th) : (int)A.ih;
    const int rmax = (int)A.ih - 1;
    ms.r_end = r_end;
    auto row_off = [&](int rr) {
        rr = rr < 0 ? 0 : (rr > rmax ? rmax : rr);
        return in_off + (uint32_t)rr * (A.iw * 4);
    };
    constexpr bool HIDDEN = NUS_LZ_ASM_LOADS != 0;
    constexpr int NL = BLEND ? 2 : 1;
    __shared__ u32x4 lds_rows[HIDDEN ? 4 : 1][HIDDEN ? kLzDepth * NL : 1][64];
    RowRing<BLEND> ring;
    {
        const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        ring.slot = lds_rows[HIDDEN ? w : 0];
        ring.lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&lds_rows[HIDDEN ? w : 0][0][0]);
        ring.lane = lane;
#if NUS_LZ_CONTIG_STORES && !NUS_LZ_SWAP_STORES
        __shared__ u32x4 lds_stage[4][128]; // one output row (2 KiB) per wave, turned round between compute and store order
        st.stage = lds_stage[w];
#else
        st.stage = nullptr;
#endif
    }

    PhaseWeights W;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        W.e[j] = vgpr(A.wxe[j]);
        W.o[j] = vgpr(A.wxo[j]);
    }
    float win[6][16];
    AlphaRun arun;
    RowRaw<BLEND> raw[2];
    {
        // the six rows of the first window (ordinary loads, all in flight together), then the first requests
        RowRaw<BLEND> first[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) first[j] = fetch_row_plain<BLEND>(src, src_b, row_off(r0 - 3 + j));
        if (HIDDEN) {
#pragma unroll
            for (int j = 0; j < kLzDepth; ++j) ring.request(j, src, src_b, row_off(r0 + 3 + j));
        } else {
            raw[0] = fetch_row_plain<BLEND>(src, src_b, row_off(r0 + 3));
            raw[1] = fetch_row_plain<BLEND>(src, src_b, row_off(r0 + 4));
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint4 px = resolve_row<BLEND>(first[j], t, A.sel);
            if (!EXACT && NUS_LZ_BLEND_OPAQUE_PATH_OK(BLEND)) alpha_run_push(arun, px);
            cvt_row(px, win[j]);
            if (j >= 3) lanczos_x2_store_mid<UNIT>(px, ms, r0 - 3 + j, A.iw * 4); // rows r0 .. r0+2 (the loop stores r0+3 ..)
        }
        // the hand-counted waits of the loop assume that nothing older than its own instructions is outstanding
        if (HIDDEN) wait_vmcnt<0, 0>();
    }
    for (int rbase = r0; rbase < r_end; rbase += 6) {
        // 6-way unrolled so the rotating window indices are compile-time constants.  The block leaves the loop
        // after its last row: a step is never skipped with a later one still to run, so every path through
        // the loop carries the vector memory instructions the hand-counted waits assume.
#define NUS_LZ_STEP(S) \
        lanczos_x2_step<EXACT, BLEND, UNIT, S, NARROW>(win, ring, raw, arun, rbase + S, in_off, st, A, W, src, src_b, rs, t, ms); \
        if (S < 5 && rbase + S + 1 >= r_end) break
        NUS_LZ_STEP(0);
        NUS_LZ_STEP(1);
        NUS_LZ_STEP(2);
        NUS_LZ_STEP(3);
        NUS_LZ_STEP(4);
        NUS_LZ_STEP(5);
#undef NUS_LZ_STEP
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
    uint32_t row0, row_end; // input rows of this launch (UpscaleLaunch::row0 / rows)
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
    const int r = (int)(A.row0 + blockIdx.x * kWave + threadIdx.x);
    if (r >= (int)A.row_end) return;
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
        raw[j][0] = resolve_row<BLEND>(fetch_row_plain<BLEND>(src, src_b, off), A.t, A.sel);
        raw[j][1] = resolve_row<BLEND>(fetch_row_plain<BLEND>(src, src_b, off + 16), A.t, A.sel);
    }
    if (side == 0)
        lanczos_x2_edge_rows<EXACT, 0>(A, raw, r, dst);
    else
        lanczos_x2_edge_rows<EXACT, 1>(A, raw, r, dst);
}

} // namespace

hipError_t launch_lanczos_x2(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave)
{
    LanczosX2Args A{};
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
    A.rb0 = 0;
    if (L.rows) { // a band of the frame: whole row blocks from row0 on (the last block of the frame may be short)
        if (L.row0 % A.th != 0 || L.row0 >= L.ih) return hipErrorInvalidValue;
        const uint32_t row_end = L.row0 + L.rows < L.ih ? L.row0 + L.rows : L.ih;
        if (row_end != L.ih && row_end % A.th != 0) return hipErrorInvalidValue;
        A.rb0 = L.row0 / A.th;
        A.nrowblocks = cdiv(row_end - L.row0, A.th);
    }
    A.in_frame_bytes = L.in_stride ? L.in_stride : (size_t)L.iw * L.ih * 4;
    A.in_b_frame_bytes = L.in_b_stride;
    A.out_frame_bytes = (size_t)L.ow * L.oh * 4;
    A.t = L.blend_t;
    A.sel = L.in_sel;
    const int blend = L.in_b == nullptr ? 0 : (L.blend_t == 0.5f ? 1 : 2);
#ifndef NUS_LZ_NARROW
#define NUS_LZ_NARROW 1 // dev macro: 0 = Catmull-Rom / Triangle through the six-tap instantiations (A/B timing)
#endif
    // (the interior vertical weights are the horizontal ones: host-checked; the border rows take their own from the table)
    const bool narrow = NUS_LZ_NARROW && blend == 0 && A.wxe[0] == 0.0f && A.wxe[5] == 0.0f && A.wxo[0] == 0.0f && A.wxo[5] == 0.0f;
    const uint32_t nwaves = A.nstrips * A.nrowblocks;
    hipError_t e = for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.in_b = L.in_b ? L.in_b + chunk_first_frame(L, in) * L.in_b_stride : nullptr;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
#define NUS_LZ(E, B) hipLaunchKernelGGL((k_lanczos3_x2<E, B, false>), grid, block, 0, L.stream, A)
        if (narrow) { // Catmull-Rom / Triangle: the outermost taps of both phases are 0 on both axes -- four taps per pass
            if (exact)
                hipLaunchKernelGGL((k_lanczos3_x2<true, 0, false, true>), grid, block, 0, L.stream, A);
            else
                hipLaunchKernelGGL((k_lanczos3_x2<false, 0, false, true>), grid, block, 0, L.stream, A);
        } else if (exact) {
            if (blend == 0) NUS_LZ(true, 0); else if (blend == 1) NUS_LZ(true, 1); else NUS_LZ(true, 2);
        } else {
            if (blend == 0) NUS_LZ(false, 0); else if (blend == 1) NUS_LZ(false, 1); else NUS_LZ(false, 2);
        }
#undef NUS_LZ
    });
    return e;
}

// One pipeline step in one launch (see k_lanczos3_x2, UNIT): L.in / L.in_b are the frames A_k / B_k of the pairs, L.out the
// up-scaled real frames, U.out_mid the up-scaled in-between frames, U.mid the in-between frames themselves (may be null).
// Follow it with launch_lanczos_x2_edges for both outputs.
hipError_t launch_lanczos_x2_unit(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave,
                                  const UnitOutputs &U)
{
    if (L.in_b == nullptr || U.out_mid == nullptr) return hipErrorInvalidValue;
    LanczosX2Args A{};
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
    A.in = L.in;
    A.in_b = L.in_b;
    A.out = L.out;
    A.out_mid = U.out_mid;
    A.mid = U.mid;
    A.nframes = L.n_frames;
    A.order = U.order;
    const uint64_t nwaves = (uint64_t)2 * A.nstrips * A.nrowblocks * L.n_frames;
    if (nwaves / 4 + 1 >= (1ull << 31)) return hipErrorInvalidValue;
    const dim3 block(256), grid((uint32_t)((nwaves + 3) / 4));
    const bool half = L.blend_t == 0.5f;
#define NUS_LZU(E, B) hipLaunchKernelGGL((k_lanczos3_x2<E, B, true>), grid, block, 0, L.stream, A)
    if (exact) {
        if (half) NUS_LZU(true, 1); else NUS_LZU(true, 2);
    } else {
        if (half) NUS_LZU(false, 1); else NUS_LZU(false, 2);
    }
#undef NUS_LZU
    return hipGetLastError();
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
    A.row0 = L.rows ? L.row0 : 0;
    A.row_end = L.rows ? (L.row0 + L.rows < L.ih ? L.row0 + L.rows : L.ih) : L.ih;
    const int blend = L.in_b == nullptr ? 0 : (L.blend_t == 0.5f ? 1 : 2);
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.in_b = L.in_b ? L.in_b + chunk_first_frame(L, in) * L.in_b_stride : nullptr;
        A.out = out;
        const dim3 block(kWave), grid(cdiv(A.row_end - A.row0, kWave), 2, n);
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

// nus_k_upscale.hip -- nearest and bilinear kernels (gfx950, wave64); the resize filters live in nus_k_resize*.hip.
//
// Reference arithmetic being reproduced (paths relative to the reference checkout):
//   nearest    nu_scaler_core/src/upscale/mod.rs:184-206 == Nu_scale/src/upscale/common.rs:188-198
//   bilinear   Nu_scale/src/upscale/common.rs:199-231 (CPU form, the oracle);
//              nu_scaler_core/src/upscale/mod.rs:209-263 (WGSL form, optional variant)
#include "nus_device.hpp"

namespace nus {

namespace {

// ---------------------------------------------------------------------------------
// Nearest
// ---------------------------------------------------------------------------------

// Any scale.  blockDim = (64, 4): each wave owns a 256-px (VEC) or 64-px column segment and walks
// `rows_per_wave` output rows; the gathered source pixels are kept in registers and re-fetched only
// when the source row changes (an upscale stores each gathered row several times).
template <bool VEC>
__global__ __launch_bounds__(256) void k_nearest_table(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const uint32_t *__restrict__ sx, const uint32_t *__restrict__ sy,
    uint32_t iw, uint32_t ow, uint32_t oh, uint32_t rows_per_wave, size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    constexpr int N = VEC ? 4 : 1;
    const uint32_t rb = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t y_begin = rb * rows_per_wave;
    const uint32_t x = (blockIdx.x * kWave + threadIdx.x) * N;
    if (y_begin >= oh || x >= ow) return;
    const uint32_t y_end = umin(y_begin + rows_per_wave, oh);
    const uint32_t *base = in + (size_t)blockIdx.z * in_frame_px;
    uint32_t *dst = out + (size_t)blockIdx.z * out_frame_px + x;
    uint32_t s[N], o[N];
    if (VEC) {
        const uint4 v = *reinterpret_cast<const uint4 *>(sx + x);
        s[0] = v.x; s[N > 1 ? 1 : 0] = v.y; s[N > 2 ? 2 : 0] = v.z; s[N > 3 ? 3 : 0] = v.w;
    } else {
        s[0] = sx[x];
    }
    uint32_t have = 0xffffffffu;
    for (uint32_t y = y_begin; y < y_end; ++y) {
        const uint32_t r = uniform_load(sy, y);
        if (r != have) { // wave-uniform
            const uint32_t *src = base + (size_t)r * iw;
#pragma unroll
            for (int i = 0; i < N; ++i) o[i] = swz(src[s[i]], sel);
            have = r;
        }
        if (VEC)
            store_out16<true>(dst + (size_t)y * ow, make_uint4(o[0], o[N > 1 ? 1 : 0], o[N > 2 ? 2 : 0], o[N > 3 ? 3 : 0]));
        else
            dst[(size_t)y * ow] = o[0];
    }
}

// Exact x2 (ow == 2*iw, oh == 2*ih, iw % 4 == 0).  A wave covers 256 input pixels of one row = 512 output pixels
// of two rows.  Lane L takes input pixels 2L, 2L+1 and 128 + 2L, 128 + 2L + 1 of that span (two 8-B loads), so
// that its two 16-B stores per output row land at 16 L and at 1024 + 16 L: every store instruction of the
// wave writes ONE CONTIGUOUS KiB.  The obvious mapping -- 4 adjacent input pixels per lane, two 16-B stores at
// a 32-B lane stride -- leaves every 128-B line half written until the second instruction arrives, which on
// gfx950 costs a write-heavy stream with reads in it a third of its rate (tools/probe_rw_mix.hip).
__global__ __launch_bounds__(256) void k_nearest_x2(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    uint32_t iw, uint32_t ih, size_t in_frame_px, size_t out_frame_px, uint32_t sel, uint32_t row0, uint32_t row_end)
{
    const uint32_t r = __builtin_amdgcn_readfirstlane(row0 + blockIdx.y * 4 + threadIdx.y);
    const uint32_t k0 = blockIdx.x * 256;
    if (r >= row_end) return; // (row_end <= ih)
    const uint32_t ow = iw * 2;
    const uint32_t *src = in + (size_t)blockIdx.z * in_frame_px + (size_t)r * iw;
    uint32_t *d = out + (size_t)blockIdx.z * out_frame_px + (size_t)(2 * r) * ow;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t k = k0 + 128 * g + 2 * threadIdx.x;
        if (k >= iw) continue;
        const uint2 p2 = *reinterpret_cast<const uint2 *>(src + k);
        const uint32_t p0 = swz(p2.x, sel), p1 = swz(p2.y, sel);
        const uint4 o = make_uint4(p0, p0, p1, p1);
        store_out16<true>(d + 2 * k, o);
        store_out16<true>(d + ow + 2 * k, o);
    }
}

// ---------------------------------------------------------------------------------
// Bilinear
// ---------------------------------------------------------------------------------

// Horizontal lerp of one source row for the lane's N outputs (common.rs:221-222; upscale/mod.rs:255-256
// for the WGSL form, whose texels are first divided by 255).
template <int N, bool WGSL>
__device__ __forceinline__ void bilinear_hrow(const uint32_t *__restrict__ row, const uint32_t (&xi)[N], const float (&xf)[N],
                                              uint32_t iw, uint32_t sel, float (&h)[N * 4])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t p0 = swz(row[xi[i]], sel), p1 = swz(row[umin(xi[i] + 1, iw - 1)], sel);
        const float dx = xf[i], ndx = 1.0f - dx;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = ch_f32(p0, c), b = ch_f32(p1, c);
            if (WGSL) {
                a = div_by_recip(a, 255.0f, 1.0f / 255.0f); // == a / 255.0f for every u8 (checked on the CPU)
                b = div_by_recip(b, 255.0f, 1.0f / 255.0f);
            }
            h[i * 4 + c] = a * ndx + b * dx;
        }
    }
}

// The same from a STAGED source row (round 5).  bilinear_hrow fetches its 2 N texels with 2 N one-dword gathers per lane; with four
// outputs per lane that is eight vector memory instructions per source row for ~300 bytes of texels per wave, and the texture
// address unit, not the arithmetic, bounds the kernel (two gathers instead of eight, as an experiment: 5.5 -> 4.1 us per frame at
// 1080p x1.1; profiles/r05_bilinear_table_staged_rows.txt).  On an up-scale the 256 outputs of a wave read at most 258 consecutive
// texels (261 from a start rounded down to a multiple of 4): the wave loads them once, 16 aligned bytes per lane, parks them in its LDS row and every lane picks its texel pairs from there
// (ds_read2_b32).  Texels past the row's end are loaded as the last texel, which is what the CPU's clamped x1 reads.  Down-scaling by
// 1.9 ... 2 takes the same road with KS = 2 wide loads per lane (a wave's outputs then reach 512 texels and need all of them).
template <int KS>
struct StagedRow {
    uint4 v[KS];    // texels s0 + 256 k + 4 lane .. + 3
    uint32_t extra; // lanes 0 .. 5: texels s0 + 256 KS .. + 5 (s0 is rounded down to a multiple of 4: aligned 16-byte loads)
};

template <int KS>
__device__ __forceinline__ StagedRow<KS> bilinear_stage_load(const uint32_t *__restrict__ row, uint32_t s0, uint32_t iw, uint32_t lane)
{
    StagedRow<KS> r;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const uint32_t t = s0 + 256 * k + 4 * lane;
        if (t + 3 < iw) {
            r.v[k] = *reinterpret_cast<const uint4 *>(row + t); // (16-byte aligned when the row is: s0 % 4 == 0)
        } else { // the row's end: clamped, one texel at a time
            r.v[k] = make_uint4(row[umin(t, iw - 1)], row[umin(t + 1, iw - 1)], row[umin(t + 2, iw - 1)], row[iw - 1]);
        }
    }
    r.extra = lane < 6 ? row[umin(s0 + 256 * KS + lane, iw - 1)] : 0u;
    return r;
}

// the lanes' texel pairs of a staged row: raw -> LDS -> 8 registers (LDS instructions of one wave execute in order: the lanes' writes
// are in place when the reads are served, and the next row's writes stay behind these reads)
struct TexelPairs {
    uint32_t p[8]; // (p0, p1) of the lane's four outputs
};

template <int KS>
__device__ __forceinline__ TexelPairs bilinear_stage_pick(const StagedRow<KS> &raw, uint32_t *__restrict__ stage, const uint32_t (&rel)[4],
                                                          uint32_t lane)
{
#pragma unroll
    for (int k = 0; k < KS; ++k) *reinterpret_cast<uint4 *>(stage + 256 * k + 4 * lane) = raw.v[k];
    if (lane < 6) stage[256 * KS + lane] = raw.extra;
    __builtin_amdgcn_wave_barrier(); // (compiler only)
    TexelPairs t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.p[2 * i] = stage[rel[i]], t.p[2 * i + 1] = stage[rel[i] + 1];
    __builtin_amdgcn_wave_barrier();
    return t;
}

template <bool WGSL>
__device__ __forceinline__ void bilinear_hrow_pairs(const TexelPairs &t, const float (&xf)[4], uint32_t sel, float (&h)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p0 = swz(t.p[2 * i], sel), p1 = swz(t.p[2 * i + 1], sel);
        const float dx = xf[i], ndx = 1.0f - dx;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = ch_f32(p0, c), b = ch_f32(p1, c);
            if (WGSL) {
                a = div_by_recip(a, 255.0f, 1.0f / 255.0f);
                b = div_by_recip(b, 255.0f, 1.0f / 255.0f);
            }
            h[i * 4 + c] = a * ndx + b * dx;
        }
    }
}

// Any scale; coordinates come from host-built tables so no division runs here and the index /
// fraction values are exactly the CPU's.  blockDim = (64, 4): each wave owns a column segment
// (4 outputs per lane when VEC) and walks `rows_per_wave` output rows.  The horizontally lerped
// source rows ("top" / "bottom" of common.rs:221-222) depend only on the source row, so they stay
// in registers while consecutive output rows map to the same source rows -- on an upscale each is
// reused for ~scale output rows -- and only the vertical lerp + pack runs per output pixel.
template <bool VEC, bool WGSL, int KS = 0>
__global__ __launch_bounds__(256) void k_bilinear_table(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const uint32_t *__restrict__ x0t, const float *__restrict__ fxt,
    const uint32_t *__restrict__ y0t, const float *__restrict__ fyt,
    uint32_t iw, uint32_t ih, uint32_t ow, uint32_t oh, uint32_t rows_per_wave, size_t in_frame_px, size_t out_frame_px,
    uint32_t sel)
{
    constexpr int N = VEC ? 4 : 1;
    constexpr bool STAGE = KS > 0;
    constexpr int KSN = STAGE ? KS : 1;
    const uint32_t rb = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t y_begin = rb * rows_per_wave;
    const uint32_t x_own = (blockIdx.x * kWave + threadIdx.x) * N;
    // (STAGE: lanes past the row's end stay -- they carry texels of the staged source row -- and only skip their stores)
    const bool live = x_own < ow;
    if (y_begin >= oh || (!STAGE && !live)) return;
    const uint32_t x = live ? x_own : ow - N;
    const uint32_t y_end = umin(y_begin + rows_per_wave, oh);
    const uint32_t *base = in + (size_t)blockIdx.z * in_frame_px;
    uint32_t *dst = out + (size_t)blockIdx.z * out_frame_px + x;
    uint32_t xi[N];
    float xf[N];
    if (VEC) {
        const uint4 a = *reinterpret_cast<const uint4 *>(x0t + x);
        const float4 f = *reinterpret_cast<const float4 *>(fxt + x);
        xi[0] = a.x; xi[N > 1 ? 1 : 0] = a.y; xi[N > 2 ? 2 : 0] = a.z; xi[N > 3 ? 3 : 0] = a.w;
        xf[0] = f.x; xf[N > 1 ? 1 : 0] = f.y; xf[N > 2 ? 2 : 0] = f.z; xf[N > 3 ? 3 : 0] = f.w;
    } else {
        xi[0] = x0t[x];
        xf[0] = fxt[x];
    }
    static_assert(!STAGE || VEC, "the staged source row serves four outputs per lane");
    // STAGE (host-checked): the wave's 256 outputs read the texels s0 .. s0 + 256 KS + 5 of a source row
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[STAGE ? 4 : 1][STAGE ? 256 * KSN + 8 : 1];
    uint32_t *stage = s_stage[STAGE ? threadIdx.y : 0];
    const uint32_t s0 = STAGE ? __builtin_amdgcn_readfirstlane(uniform_load(x0t, (size_t)blockIdx.x * kWave * N)) & ~3u : 0u;
    uint32_t rel[4] = {0u, 0u, 0u, 0u};
    if (STAGE) {
#pragma unroll
        for (int i = 0; i < N; ++i) rel[i] = xi[i] - s0;
    }
    // STAGE: a two-deep pipeline over the source rows (an up-scale consumes them one by one).  When row r has been lerped, row r + 1
    // -- requested a step earlier -- goes through LDS into the lanes' texel-pair registers, and row r + 2 is requested: the global
    // load's latency and the LDS round trip are both spent under the vertical lerps of the output rows in between.  A row that is in
    // neither stage (a block's first row) is fetched on the spot.
    StagedRow<KSN> pre = {};
    TexelPairs tex = {{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
    uint32_t pre_row = 0xffffffffu, tex_row = 0xffffffffu; // wave-uniform
    auto hrow = [&](uint32_t r, float (&h)[N * 4]) __attribute__((always_inline)) {
        if constexpr (STAGE) {
            float h16[16];
            if (r != tex_row) {
                const StagedRow<KSN> raw = r == pre_row ? pre : bilinear_stage_load<KSN>(base + (size_t)r * iw, s0, iw, threadIdx.x);
                tex = bilinear_stage_pick<KSN>(raw, stage, rel, threadIdx.x);
            }
            bilinear_hrow_pairs<WGSL>(tex, reinterpret_cast<const float (&)[4]>(xf), sel, h16);
#pragma unroll
            for (int k = 0; k < N * 4; ++k) h[k] = h16[k];
            const uint32_t r1 = umin(r + 1, ih - 1), r2 = umin(r + 2, ih - 1);
            if (pre_row != r1) pre = bilinear_stage_load<KSN>(base + (size_t)r1 * iw, s0, iw, threadIdx.x); // (a block's first rows, rows skipped on a down-scale: waited for here)
            tex = bilinear_stage_pick<KSN>(pre, stage, rel, threadIdx.x);
            tex_row = r1;
            pre = bilinear_stage_load<KSN>(base + (size_t)r2 * iw, s0, iw, threadIdx.x);
            pre_row = r2;
        } else {
            bilinear_hrow<N, WGSL>(base + (size_t)r * iw, xi, xf, iw, sel, h);
        }
    };
    float ht[N * 4], hb[N * 4]; // lerped source rows top_row / bot_row
    uint32_t top_row = 0xffffffffu, bot_row = 0xffffffffu;
    for (uint32_t y = y_begin; y < y_end; ++y) {
        const uint32_t y0 = uniform_load(y0t, y);
        const uint32_t y1 = umin(y0 + 1, ih - 1);
        float dy = uniform_load(fyt, y);
        asm volatile("" : "+v"(dy)); // VGPR copy: scalar operands halve the VALU issue rate
        const float ndy = 1.0f - dy;
        if (y0 != top_row) { // wave-uniform
            if (y0 == bot_row) {
#pragma unroll
                for (int k = 0; k < N * 4; ++k) ht[k] = hb[k];
            } else {
                hrow(y0, ht);
            }
            top_row = y0;
        }
        if (y1 != bot_row) {
            if (y1 == y0) {
#pragma unroll
                for (int k = 0; k < N * 4; ++k) hb[k] = ht[k];
            } else {
                hrow(y1, hb);
            }
            bot_row = y1;
        }
        uint32_t o[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint32_t px = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = ht[i * 4 + c] * ndy + hb[i * 4 + c] * dy;
                // CPU form: clamp(0,255) as u8; WGSL form: u32(clamp(v,0,1)*255) -- the saturating pack clamps below
                px = pack_trunc_u8(WGSL ? fminf(v, 1.0f) * 255.0f : v, c, px);
            }
            o[i] = px;
        }
        if (STAGE && !live) continue;
        if (VEC)
            store_out16<false>(dst + (size_t)y * ow, make_uint4(o[0], o[N > 1 ? 1 : 0], o[N > 2 ? 2 : 0], o[N > 3 ? 3 : 0]));
        else
            dst[(size_t)y * ow] = o[0];
    }
}

// floor((a+b)/2) per byte: v_lerp_u8 with rounding bits 0.
__device__ __forceinline__ uint32_t avg2_u8x4(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_lerp(a, b, 0u);
}

// floor((a+b+c+d)/4) per byte, exact: with l1 = floor((a+b)/2), l2 = floor((c+d)/2) the
// lost half-units are the low bits of a^b and c^d; both set adds one unit to l1+l2.
__device__ __forceinline__ uint32_t avg4_u8x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t l1 = __builtin_amdgcn_lerp(a, b, 0u);
    const uint32_t l2 = __builtin_amdgcn_lerp(c, d, 0u);
    return __builtin_amdgcn_lerp(l1, l2, (a ^ b) & (c ^ d));
}

// Exact x2, CPU arithmetic.  At x2 the fractions are 0 or 0.5, every product and sum
// of common.rs:221-226 is exact in f32 and the truncation is a floor of a quarter
// multiple, so the result equals these packed-u8 integer averages byte for byte
// (requires (ow-1)*iw < 2^24 so that x*iw/ow is exact; checked by the host).
// (The two-groups-per-lane mapping of k_nearest_x2, which makes every store instruction one contiguous KiB, was measured
// here too -- with 8-B + 4-B loads and with one 12-B buffer load per group and row -- and lost 4-7 % to this form on two
// boxes: profiles/r02_x2_store_shape_ab.txt.  These waves live for one row; the half-line stores that cost the
// row-walking resize kernel a third of its rate do not cost them.)
// Each lane: 4 input pixels of rows r and r+1 -> 8x2 output pixels.
__global__ __launch_bounds__(256) void k_bilinear_x2_int(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    uint32_t iw, uint32_t ih, size_t in_frame_px, size_t out_frame_px, uint32_t sel, uint32_t row0, uint32_t row_end)
{
    const uint32_t r = __builtin_amdgcn_readfirstlane(row0 + blockIdx.y * 4 + threadIdx.y);
    const uint32_t k = (blockIdx.x * kWave + threadIdx.x) * 4;
    if (r >= row_end || k >= iw) return; // (row_end <= ih; row r + 1 is read from the frame, clamped at ITS last row)
    const uint32_t ow = iw * 2;
    const uint32_t r1 = umin(r + 1, ih - 1);
    const uint32_t k4 = umin(k + 4, iw - 1);
    const uint32_t *base = in + (size_t)blockIdx.z * in_frame_px;
    const uint32_t *rowp = base + (size_t)r * iw;
    const uint32_t *rowq = base + (size_t)r1 * iw;
    const uint4 pv = swz4(*reinterpret_cast<const uint4 *>(rowp + k), sel);
    const uint4 qv = swz4(*reinterpret_cast<const uint4 *>(rowq + k), sel);
    const uint32_t p[5] = {pv.x, pv.y, pv.z, pv.w, swz(rowp[k4], sel)};
    const uint32_t q[5] = {qv.x, qv.y, qv.z, qv.w, swz(rowq[k4], sel)};
    uint32_t top[8], bot[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        top[2 * i] = p[i];
        top[2 * i + 1] = avg2_u8x4(p[i], p[i + 1]);
        bot[2 * i] = avg2_u8x4(p[i], q[i]);
        bot[2 * i + 1] = avg4_u8x4(p[i], p[i + 1], q[i], q[i + 1]);
    }
    uint32_t *d = out + (size_t)blockIdx.z * out_frame_px + (size_t)(2 * r) * ow + 2 * k;
    store_out16<false>(d, make_uint4(top[0], top[1], top[2], top[3]));
    store_out16<false>(d + 4, make_uint4(top[4], top[5], top[6], top[7]));
    store_out16<false>(d + ow, make_uint4(bot[0], bot[1], bot[2], bot[3]));
    store_out16<false>(d + ow + 4, make_uint4(bot[4], bot[5], bot[6], bot[7]));
}

} // namespace

// ---- Nearest and bilinear (CPU form) at the small rational factors P/Q (3/2, 4/3, 3, 4; 2 where the x2 kernels' --------
// ---- width % 4 == 0 does not hold) ------------------------------------------------------------------------------------
// Output o = P g + p samples the input group g = (Q g .. Q g + Q - 1): source index Q g + p Q / P (integer division) --
// host-checked on the tables for every output.  3/2 is the factor the reference's benchmark entry points default to
// (nu_scaler_py/nu_scaler/benchmark.py:63); nearest is what every technology but Wgpu falls back to
// (upscale/mod.rs:102-115), bilinear what its AMD path does (gpu/detector.rs:181-190).
// A lane owns one group of columns: Q pixels in (plus the next one for the bilinear pairs, clamped at the row's end as the
// CPU clamps x1), P pixels = 4 P contiguous bytes out; rows likewise in groups of Q, P output rows each.
template <int P, int Q>
struct RatioPx {
    uint32_t px[Q + 1];
};

template <int P, int Q>
__device__ __forceinline__ RatioPx<P, Q> ratio_fetch(const uint32_t *row, uint32_t g, uint32_t iw)
{
    RatioPx<P, Q> t;
    const uint32_t *p = row + Q * g;
    if constexpr (Q == 2) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        t.px[0] = v.x, t.px[1] = v.y;
    } else {
#pragma unroll
        for (int k = 0; k < Q; ++k) t.px[k] = p[k];
    }
    t.px[Q] = row[umin(Q * g + Q, iw - 1)];
    return t;
}

template <int P>
__device__ __forceinline__ void ratio_store(uint32_t *d, const uint32_t (&o)[P])
{
    if constexpr (P == 4) {
        store_out16<true>(d, make_uint4(o[0], o[1], o[2], o[3]));
    } else {
#pragma unroll
        for (int k = 0; k < P; ++k) d[k] = o[k]; // (merged into one global_store_dwordx3)
    }
}

template <int P, int Q>
__global__ __launch_bounds__(256) void k_nearest_ratio(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t iw,
                                                       uint32_t ih, uint32_t ow, size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    const uint32_t g = blockIdx.x * kWave + threadIdx.x;                              // input column group
    const uint32_t m = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y); // input row group
    if (Q * g >= iw || Q * m >= ih) return;
    const uint32_t *src = in + (size_t)blockIdx.z * in_frame_px + (size_t)(Q * m) * iw;
    uint32_t *d = out + (size_t)blockIdx.z * out_frame_px + (size_t)(P * m) * ow + P * g;
    uint32_t o[Q][P];
#pragma unroll
    for (int r = 0; r < Q; ++r) {
        const RatioPx<P, Q> t = ratio_fetch<P, Q>(src + (size_t)r * iw, g, iw);
#pragma unroll
        for (int p = 0; p < P; ++p) o[r][p] = swz(t.px[p * Q / P], sel);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) ratio_store<P>(d + (size_t)p * ow, o[p * Q / P]);
}

// Bilinear: the fractions of the phases that do not land on a pixel are rounded in f32 and come per lane / per row from
// the tables; a phase with p Q % P == 0 lands on a pixel, its fraction is 0 (host-checked) and its lerp
// p00 * 1 + p10 * 0 is p00 itself.  The horizontally lerped rows of a row group stay in registers.  Same expressions as
// k_bilinear_table (common.rs:221-227) otherwise; at 3/2 about 0.6 of its instructions per pixel.
template <int P, int Q>
__global__ __launch_bounds__(256) void k_bilinear_ratio(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out, const float *__restrict__ fxt, const float *__restrict__ fyt,
    uint32_t iw, uint32_t ih, uint32_t ow, uint32_t groups_per_wave, size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    const uint32_t g = blockIdx.x * kWave + threadIdx.x; // this lane's input column group
    const uint32_t rb = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t m_begin = rb * groups_per_wave, m_end = umin(m_begin + groups_per_wave, ih / Q);
    if (m_begin >= m_end || Q * g >= iw) return;
    const uint32_t *base = in + (size_t)blockIdx.z * in_frame_px;
    uint32_t *dst = out + (size_t)blockIdx.z * out_frame_px + P * g;
    float dx[P], ndx[P];
#pragma unroll
    for (int p = 0; p < P; ++p) dx[p] = fxt[P * g + p], ndx[p] = 1.0f - dx[p];
    auto fetch = [&](uint32_t r) { return ratio_fetch<P, Q>(base + (size_t)umin(r, ih - 1) * iw, g, iw); };
    // the lane's P horizontally lerped values of one source row: h[4 p + c]
    auto hrow = [&](const RatioPx<P, Q> &t, float (&h)[4 * P]) {
        float f[Q + 1][4];
#pragma unroll
        for (int k = 0; k <= Q; ++k) {
            const uint32_t px = swz(t.px[k], sel);
#pragma unroll
            for (int c = 0; c < 4; ++c) f[k][c] = ch_f32(px, c);
        }
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                h[4 * p + c] = (p * Q % P == 0) ? f[p * Q / P][c]                                         // a * (1 - 0) + b * 0
                                                : f[p * Q / P][c] * ndx[p] + f[p * Q / P + 1][c] * dx[p]; // top = p00 * (1 - dx) + p10 * dx
    };
    auto emit = [&](uint32_t y, const float (&v)[4 * P]) {
        uint32_t o[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            uint32_t px = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) px = pack_trunc_u8(v[4 * p + c], c, px); // clamp(0, 255) as u8
            o[p] = px;
        }
        ratio_store<P>(dst + (size_t)y * ow, o);
    };
    float h[Q + 1][4 * P]; // source rows Q m .. Q m + Q
    hrow(fetch(Q * m_begin), h[0]);
    RatioPx<P, Q> nxt[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) nxt[k] = fetch(Q * m_begin + 1 + k);
    for (uint32_t m = m_begin; m < m_end; ++m) {
#pragma unroll
        for (int k = 0; k < Q; ++k) hrow(nxt[k], h[1 + k]);
        if (m + 1 < m_end) { // the next group's rows: in flight during this one
#pragma unroll
            for (int k = 0; k < Q; ++k) nxt[k] = fetch(Q * (m + 1) + 1 + k);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (p * Q % P == 0) {
                emit(P * m + p, h[p * Q / P]); // top * (1 - 0) + bottom * 0
            } else {
                float dy = uniform_load(fyt, (size_t)(P * m + p));
                asm volatile("" : "+v"(dy)); // VGPR copy: scalar operands halve the VALU issue rate
                const float ndy = 1.0f - dy;
                float v[4 * P];
#pragma unroll
                for (int k = 0; k < 4 * P; ++k) v[k] = h[p * Q / P][k] * ndy + h[p * Q / P + 1][k] * dy;
                emit(P * m + p, v);
            }
        }
#pragma unroll
        for (int k = 0; k < 4 * P; ++k) h[0][k] = h[Q][k];
    }
}

const char *variant_name(Variant v){
    switch (v) {
    case Variant::NearestTable: return "nearest_table";
    case Variant::NearestX2: return "nearest_x2_vec16";
    case Variant::NearestRatio: return "nearest_ratio";
    case Variant::BilinearTable: return "bilinear_table_f32";
    case Variant::BilinearX2Int: return "bilinear_x2_packed_u8";
    case Variant::BilinearRatio: return "bilinear_ratio_f32";
    case Variant::LanczosGeneral: return "lanczos3_general";
    case Variant::ResizeRows: return "resize_rows_lds";
    case Variant::ResizeWin: return "resize_regwin_lds";
    case Variant::ResizeDown: return "resize_down_stream";
    case Variant::LanczosX2RegWin: return "lanczos3_x2_regwin";
    case Variant::LanczosXsRegWin: return "lanczos3_xs_regwin";
    case Variant::LanczosR32RegWin: return "lanczos3_r32_regwin";
    case Variant::LanczosR43RegWin: return "lanczos3_r43_regwin";
    case Variant::LanczosPqRegWin: return "lanczos3_pq_regwin";
    case Variant::FsrEasu: return "fsr1_easu_tile";
    case Variant::FsrRcas: return "fsr1_rcas_rows";
    case Variant::Fsr1Fused: return "fsr1_easu_rcas_fused_lds";
    case Variant::Fsr1TwoPass: return "fsr1_easu_then_rcas_rows";
    }
    return "?";
}

namespace {
// Output rows per wave of the row-walking table kernels: tall enough that the lerped / gathered source
// rows get reused, small enough that the launch still has a few thousand waves.
uint32_t rows_per_wave_for(const UpscaleLaunch &L, uint32_t cols_per_wave, uint32_t n_frames)
{
    const uint64_t strips = cdiv(L.ow, cols_per_wave);
    const uint64_t t = (uint64_t)L.oh * strips * n_frames / 8192;
    return (uint32_t)(t < 8 ? 8 : (t > 64 ? 64 : t));
}
} // namespace

hipError_t launch_nearest_table(const UpscaleLaunch &L, const DeviceTables &T)
{
    const bool vec = (L.ow % 4) == 0;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint32_t rpw = rows_per_wave_for(L, vec ? 256 : 64, n);
        const dim3 block(kWave, 4), grid(cdiv(L.ow, vec ? 256 : 64), cdiv(cdiv(L.oh, rpw), 4), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
        if (vec)
            hipLaunchKernelGGL(k_nearest_table<true>, grid, block, 0, L.stream, i32, o32, T.nn_sx, T.nn_sy, L.iw, L.ow, L.oh, rpw, ipx, opx, L.in_sel);
        else
            hipLaunchKernelGGL(k_nearest_table<false>, grid, block, 0, L.stream, i32, o32, T.nn_sx, T.nn_sy, L.iw, L.ow, L.oh, rpw, ipx, opx, L.in_sel);
    });
}

// (P, Q) of the fixed-ratio nearest / bilinear kernels for this launch, or false
static bool ratio_of(const UpscaleLaunch &L, uint32_t &P, uint32_t &Q)
{
    // ({7, 5}: nearest only -- the bilinear kernel's Q + 1 lerped rows of 4 P values do not fit the registers there; the host asks accordingly)
    static const uint32_t kRatios[][2] = {{3, 2}, {4, 3}, {3, 1}, {4, 1}, {2, 1}, {5, 4}, {6, 5}, {5, 3}, {5, 2}, {7, 2}, {7, 5}};
    for (const auto &r : kRatios)
        if ((uint64_t)L.ow * r[1] == (uint64_t)L.iw * r[0] && (uint64_t)L.oh * r[1] == (uint64_t)L.ih * r[0] && L.iw % r[1] == 0 &&
            L.ih % r[1] == 0) {
            P = r[0], Q = r[1];
            return true;
        }
    return false;
}

#define NUS_RATIO_DISPATCH(CALL)                 \
    if (P == 3 && Q == 2) { CALL(3, 2); }        \
    else if (P == 4 && Q == 3) { CALL(4, 3); }   \
    else if (P == 3 && Q == 1) { CALL(3, 1); }   \
    else if (P == 4 && Q == 1) { CALL(4, 1); }   \
    else if (P == 5 && Q == 4) { CALL(5, 4); }   \
    else if (P == 6 && Q == 5) { CALL(6, 5); }   \
    else if (P == 5 && Q == 3) { CALL(5, 3); }   \
    else if (P == 5 && Q == 2) { CALL(5, 2); }   \
    else if (P == 7 && Q == 2) { CALL(7, 2); }   \
    else { CALL(2, 1); }

hipError_t launch_nearest_ratio(const UpscaleLaunch &L)
{
    uint32_t P = 0, Q = 0;
    if (!ratio_of(L, P, Q)) return hipErrorInvalidValue;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const dim3 block(kWave, 4), grid(cdiv(L.iw / Q, kWave), cdiv(L.ih / Q, 4), n);
#define NUS_NR(PP, QQ)                                                                                                          \
    hipLaunchKernelGGL((k_nearest_ratio<PP, QQ>), grid, block, 0, L.stream, reinterpret_cast<const uint32_t *>(in),             \
                       reinterpret_cast<uint32_t *>(out), L.iw, L.ih, L.ow, ipx, opx, L.in_sel)
        if (P == 7 && Q == 5) { NUS_NR(7, 5); }
        else NUS_RATIO_DISPATCH(NUS_NR)
#undef NUS_NR
    });
}

hipError_t launch_nearest_x2(const UpscaleLaunch &L)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint32_t r0 = L.rows ? L.row0 : 0, r1 = L.rows ? (L.row0 + L.rows < L.ih ? L.row0 + L.rows : L.ih) : L.ih;
        const dim3 block(kWave, 4), grid(cdiv(L.iw, 256), cdiv(r1 - r0, 4), n);
        hipLaunchKernelGGL(k_nearest_x2, grid, block, 0, L.stream, reinterpret_cast<const uint32_t *>(in),
                           reinterpret_cast<uint32_t *>(out), L.iw, L.ih, ipx, opx, L.in_sel, r0, r1);
    });
}

hipError_t launch_bilinear_table(const UpscaleLaunch &L, const DeviceTables &T, bool wgsl_form)
{
    const bool vec = (L.ow % 4) == 0;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint32_t rpw = rows_per_wave_for(L, vec ? 256 : 64, n);
        const dim3 block(kWave, 4), grid(cdiv(L.ow, vec ? 256 : 64), cdiv(cdiv(L.oh, rpw), 4), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
#define NUS_BL(V, W, S)                                                                                         \
    hipLaunchKernelGGL((k_bilinear_table<V, W, S>), grid, block, 0, L.stream, i32, o32, T.bl_x0, T.bl_fx, T.bl_y0, \
                       T.bl_fy, L.iw, L.ih, L.ow, L.oh, rpw, ipx, opx, L.in_sel)
        // the source rows staged through LDS where a wave's 256 outputs reach at most 256 KS texels (+ 5) AND need nearly all of them:
        // every up-scale (KS = 1) and down-scaling by 1.9 ... 2 (KS = 2; 4K -> 1080p 17.0 -> 10.8 us per frame).  Between them the
        // second wide load is half wasted (1440p -> 1080p 7.1 -> 8.0: gathers kept), beyond 2 bilinear skips texels and whole rows,
        // which the gathers never fetch (4K -> 720p 5.8 against 10.1 staged): profiles/r05_bilinear_table_staged_rows.txt
        const uint32_t reach = (uint32_t)((uint64_t)255 * L.iw / L.ow) + 1; // (+ 1: the table's f32 indices may sit one above the quotient)
        const int ks = !(vec && L.iw >= 4) ? 0 : (reach <= 256 ? 1 : (reach >= 480 && reach <= 512 ? 2 : 0));
        if (ks == 1) { if (wgsl_form) NUS_BL(true, true, 1); else NUS_BL(true, false, 1); }
        else if (ks == 2) { if (wgsl_form) NUS_BL(true, true, 2); else NUS_BL(true, false, 2); }
        else if (vec && wgsl_form) NUS_BL(true, true, 0);
        else if (vec) NUS_BL(true, false, 0);
        else if (wgsl_form) NUS_BL(false, true, 0);
        else NUS_BL(false, false, 0);
#undef NUS_BL
    });
}

hipError_t launch_bilinear_ratio(const UpscaleLaunch &L, const DeviceTables &T)
{
    uint32_t P = 0, Q = 0;
    if (!ratio_of(L, P, Q) || (Q == 5 && P >= 7)) return hipErrorInvalidValue;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint32_t strips = cdiv(L.iw / Q, kWave);
        uint64_t gpw = (uint64_t)(L.ih / Q) * strips * n / 8192; // row groups per wave: a few thousand waves per launch
        gpw = gpw < 4 ? 4 : (gpw > 32 ? 32 : gpw);
        const dim3 block(kWave, 4), grid(strips, cdiv(cdiv(L.ih / Q, (uint32_t)gpw), 4), n);
#define NUS_BR(PP, QQ)                                                                                                          \
    hipLaunchKernelGGL((k_bilinear_ratio<PP, QQ>), grid, block, 0, L.stream, reinterpret_cast<const uint32_t *>(in),            \
                       reinterpret_cast<uint32_t *>(out), T.bl_fx, T.bl_fy, L.iw, L.ih, L.ow, (uint32_t)gpw, ipx, opx, L.in_sel)
        NUS_RATIO_DISPATCH(NUS_BR)
#undef NUS_BR
    });
}

hipError_t launch_bilinear_x2_int(const UpscaleLaunch &L)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint32_t r0 = L.rows ? L.row0 : 0, r1 = L.rows ? (L.row0 + L.rows < L.ih ? L.row0 + L.rows : L.ih) : L.ih;
        const dim3 block(kWave, 4), grid(cdiv(L.iw, 256), cdiv(r1 - r0, 4), n);
        hipLaunchKernelGGL(k_bilinear_x2_int, grid, block, 0, L.stream, reinterpret_cast<const uint32_t *>(in),
                           reinterpret_cast<uint32_t *>(out), L.iw, L.ih, ipx, opx, L.in_sel, r0, r1);
    });
}

} // namespace nus

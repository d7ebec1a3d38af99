// nus_kernels.hip -- gfx950 (CDNA4, wave64) kernels for the NU_Scaler hot path:
// nearest / bilinear / Lanczos-3 upscaling and two-frame warp + blend on RGBA8 frames.
//
// Reference arithmetic being reproduced (paths relative to the reference checkout):
//   nearest    nu_scaler_core/src/upscale/mod.rs:184-206 == Nu_scale/src/upscale/common.rs:188-198
//   bilinear   Nu_scale/src/upscale/common.rs:199-231 (CPU form, the oracle);
//              nu_scaler_core/src/upscale/mod.rs:209-263 (WGSL form, optional variant)
//   lanczos3   image-0.24.9 imageops::resize as called at Nu_scale/src/upscale/common.rs:243-251
//   warp+blend nu_scaler_core/src/shaders/warp_blend.wgsl:18-47 (geometry),
//              nu_scaler_core/src/interpolation/mod.rs:386-411, :467-510 (rounding)
//
// All f32 arithmetic that must match the CPU oracle bit for bit is written with
// separate multiplies and adds; the whole file is compiled with -ffp-contract=off and
// fused multiply-adds appear only where spelled __builtin_fmaf (Lanczos FMA mode).
// No MFMA: no stage is a dense contraction.  Pixels are moved as one u32 each,
// 16 bytes per lane per access wherever alignment allows.
#include "nus_kernels.hpp"

#include <cstdlib>

#pragma clang fp contract(off)

// cache-policy bits of the output stores (0 = default, 2 = nt); tuning knob
#ifndef NUS_STORE_AUX
#define NUS_STORE_AUX 0
#endif

namespace nus {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ float ch_f32(uint32_t p, int c)
{
    return (float)((p >> (8 * c)) & 0xffu); // v_cvt_f32_ubyteN
}

// Rust `clamp(0,255) as u8` / `as u8` (truncate toward zero, saturate, NaN -> 0), inserted as byte c
// of acc in two
// instructions: v_floor_f32 makes the argument an integer (for negative inputs floor and trunc differ
// but both saturate to 0), so v_cvt_pk_u8_f32's rounding mode no longer matters.
__device__ __forceinline__ uint32_t pack_trunc_u8(float v, int c, uint32_t acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(floorf(v), c, acc);
}

// f32::round (half away from zero) of a value clamped to [0,255].
__device__ __forceinline__ uint32_t round_u8_exact(float v)
{
    float c = fminf(fmaxf(v, 0.0f), 255.0f);
    float r = truncf(c);
    if (c - r >= 0.5f) r += 1.0f;
    return (uint32_t)r;
}

// FMA mode: v_cvt_pk_u8_f32 converts with round-to-nearest-EVEN and saturates to
// [0,255] in one instruction (measured on gfx950: 0.5->0, 1.5->2, 2.5->2, 254.5->254,
// -1->0, 256->255; tools/probe.hip).  It differs from f32::round only on exact .5
// ties, well inside the +-1 LSB contract of this mode.
__device__ __forceinline__ uint32_t pack_u8_rne(float v, int c, uint32_t acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(v, c, acc);
}

template <bool EXACT>
__device__ __forceinline__ float mac(float acc, float v, float w)
{
    if (EXACT) return acc + v * w;   // two roundings, as the CPU restatement
    return __builtin_fmaf(v, w, acc); // one rounding
}

// Insert round(clamp(v)) as byte c of acc.
template <bool EXACT>
__device__ __forceinline__ uint32_t pack_u8(float v, int c, uint32_t acc)
{
    if (EXACT) return acc | (round_u8_exact(v) << (8 * c));
    return pack_u8_rne(v, c, acc);
}

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Input channel order.  Captured frames arrive as BGRA and the reference swizzles them on the CPU before
// upscaling (nu_scaler_core/src/lib.rs:251-270); here every kernel passes the pixels it loads through one
// v_perm_b32 whose selector `sel` is kSelRGBA (identity) or kSelBGRA (bytes 2,1,0,3), so a BGRA source
// costs no extra pass over the frame (UpscaleLaunch::in_sel).
#ifndef NUS_SWZ_ON_LOAD
#define NUS_SWZ_ON_LOAD 1 // dev macro: 0 builds the loads without the v_perm_b32 (A/B timing of its cost only)
#endif
__device__ __forceinline__ uint32_t swz(uint32_t p, uint32_t sel)
{
#if NUS_SWZ_ON_LOAD
    return __builtin_amdgcn_perm(p, p, sel);
#else
    return p;
#endif
}
__device__ __forceinline__ uint4 swz4(const uint4 v, uint32_t sel)
{
    return make_uint4(swz(v.x, sel), swz(v.y, sel), swz(v.z, sel), swz(v.w, sel));
}

__device__ __forceinline__ float div_by_recip(float x, float y, float z); // defined with the flow kernels

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
        const uint32_t r = __builtin_amdgcn_readfirstlane(sy[y]);
        if (r != have) { // wave-uniform
            const uint32_t *src = base + (size_t)r * iw;
#pragma unroll
            for (int i = 0; i < N; ++i) o[i] = swz(src[s[i]], sel);
            have = r;
        }
        if (VEC)
            *reinterpret_cast<uint4 *>(dst + (size_t)y * ow) = make_uint4(o[0], o[N > 1 ? 1 : 0], o[N > 2 ? 2 : 0], o[N > 3 ? 3 : 0]);
        else
            dst[(size_t)y * ow] = o[0];
    }
}

// Exact x2 (ow == 2*iw, oh == 2*ih, iw % 4 == 0): each lane reads 4 input pixels
// (16 B) and writes the 2x2 replication as four 16-B stores.
__global__ __launch_bounds__(256) void k_nearest_x2(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    uint32_t iw, uint32_t ih, size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    const uint32_t r = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t k = (blockIdx.x * kWave + threadIdx.x) * 4;
    if (r >= ih || k >= iw) return;
    const uint32_t ow = iw * 2;
    const uint4 p = swz4(*reinterpret_cast<const uint4 *>(in + (size_t)blockIdx.z * in_frame_px + (size_t)r * iw + k), sel);
    const uint4 o0 = make_uint4(p.x, p.x, p.y, p.y);
    const uint4 o1 = make_uint4(p.z, p.z, p.w, p.w);
    uint32_t *d = out + (size_t)blockIdx.z * out_frame_px + (size_t)(2 * r) * ow + 2 * k;
    *reinterpret_cast<uint4 *>(d) = o0;
    *reinterpret_cast<uint4 *>(d + 4) = o1;
    *reinterpret_cast<uint4 *>(d + ow) = o0;
    *reinterpret_cast<uint4 *>(d + ow + 4) = o1;
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

// Any scale; coordinates come from host-built tables so no division runs here and the index /
// fraction values are exactly the CPU's.  blockDim = (64, 4): each wave owns a column segment
// (4 outputs per lane when VEC) and walks `rows_per_wave` output rows.  The horizontally lerped
// source rows ("top" / "bottom" of common.rs:221-222) depend only on the source row, so they stay
// in registers while consecutive output rows map to the same source rows -- on an upscale each is
// reused for ~scale output rows -- and only the vertical lerp + pack runs per output pixel.
template <bool VEC, bool WGSL>
__global__ __launch_bounds__(256) void k_bilinear_table(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    const uint32_t *__restrict__ x0t, const float *__restrict__ fxt,
    const uint32_t *__restrict__ y0t, const float *__restrict__ fyt,
    uint32_t iw, uint32_t ih, uint32_t ow, uint32_t oh, uint32_t rows_per_wave, size_t in_frame_px, size_t out_frame_px,
    uint32_t sel)
{
    constexpr int N = VEC ? 4 : 1;
    const uint32_t rb = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t y_begin = rb * rows_per_wave;
    const uint32_t x = (blockIdx.x * kWave + threadIdx.x) * N;
    if (y_begin >= oh || x >= ow) return;
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
    float ht[N * 4], hb[N * 4]; // lerped source rows top_row / bot_row
    uint32_t top_row = 0xffffffffu, bot_row = 0xffffffffu;
    for (uint32_t y = y_begin; y < y_end; ++y) {
        const uint32_t y0 = __builtin_amdgcn_readfirstlane(y0t[y]);
        const uint32_t y1 = umin(y0 + 1, ih - 1);
        float dy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(fyt[y])));
        asm volatile("" : "+v"(dy)); // VGPR copy: scalar operands halve the VALU issue rate
        const float ndy = 1.0f - dy;
        if (y0 != top_row) { // wave-uniform
            if (y0 == bot_row) {
#pragma unroll
                for (int k = 0; k < N * 4; ++k) ht[k] = hb[k];
            } else {
                bilinear_hrow<N, WGSL>(base + (size_t)y0 * iw, xi, xf, iw, sel, ht);
            }
            top_row = y0;
        }
        if (y1 != bot_row) {
            if (y1 == y0) {
#pragma unroll
                for (int k = 0; k < N * 4; ++k) hb[k] = ht[k];
            } else {
                bilinear_hrow<N, WGSL>(base + (size_t)y1 * iw, xi, xf, iw, sel, hb);
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
        if (VEC)
            *reinterpret_cast<uint4 *>(dst + (size_t)y * ow) = make_uint4(o[0], o[N > 1 ? 1 : 0], o[N > 2 ? 2 : 0], o[N > 3 ? 3 : 0]);
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
// Each lane: 4 input pixels of rows r and r+1 -> 8x2 output pixels.
__global__ __launch_bounds__(256) void k_bilinear_x2_int(
    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
    uint32_t iw, uint32_t ih, size_t in_frame_px, size_t out_frame_px, uint32_t sel)
{
    const uint32_t r = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t k = (blockIdx.x * kWave + threadIdx.x) * 4;
    if (r >= ih || k >= iw) return;
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
    *reinterpret_cast<uint4 *>(d) = make_uint4(top[0], top[1], top[2], top[3]);
    *reinterpret_cast<uint4 *>(d + 4) = make_uint4(top[4], top[5], top[6], top[7]);
    *reinterpret_cast<uint4 *>(d + ow) = make_uint4(bot[0], bot[1], bot[2], bot[3]);
    *reinterpret_cast<uint4 *>(d + ow + 4) = make_uint4(bot[4], bot[5], bot[6], bot[7]);
}

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
template <bool EXACT, bool VEC, bool SMALL>
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
    float4 *s_v = reinterpret_cast<float4 *>(smem) + (size_t)threadIdx.y * (ncols_max + 8);
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

    if (threadIdx.x < 8) s_v[(cmax - cmin) + threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // slack, see the H pass
    // horizontal windows of this lane's outputs
    int32_t hl[N];
    uint32_t hn[N];
    float hw[N][8];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t xo = lane_active ? x + i : 0;
        hl[i] = lxt[xo] - cmin;
        hn[i] = nxt[xo];
        if (SMALL) {
#pragma unroll
            for (int k = 0; k < 8; ++k) hw[i][k] = wxt[(size_t)xo * stride + k]; // zero padded beyond hn
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
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
                if (SMALL) {
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
__device__ __forceinline__ uint32_t blend_px(uint32_t a, uint32_t b, float t, float nt);

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
        float acc = EXACT ? win[BASE % 6][k] * w[0] : __builtin_fmaf(win[BASE % 6][k], w[0], 0.0f);
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
        float acc = EXACT ? win[BASE % 6][k] * w[0] : __builtin_fmaf(win[BASE % 6][k], w[0], 0.0f);
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
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave >= A.nstrips * A.nrowblocks) return;
    const uint32_t strip = wave % A.nstrips;
    const uint32_t rb = wave / A.nstrips;
    const int c = (int)(strip * kLanczosX2StripCols) - 4 + lane * 4; // first input column of this lane
    int cl = c < 0 ? 0 : c;
    cl = cl > (int)A.iw - 4 ? (int)A.iw - 4 : cl;
    const bool do_store = lane >= 1 && lane <= 62 && c >= 4 && c + 8 <= (int)A.iw;
    const uint8_t *src = A.in + (size_t)blockIdx.y * A.in_frame_bytes;
    const uint8_t *src_b = BLEND ? A.in_b + (size_t)blockIdx.y * A.in_b_frame_bytes : src;
    // one buffer resource per output frame (< 2 GiB, checked by the host); non-storing lanes
    // sit at offset 2^31, outside num_records for every row
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        A.out + (size_t)blockIdx.y * A.out_frame_bytes, 0, (uint32_t)A.out_frame_bytes, 0x00020000);
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

// ---------------------------------------------------------------------------------
// Warp + blend
// ---------------------------------------------------------------------------------

// Zero flow (the live reference behaviour, wgpu_interpolator.rs:275-295): sample
// positions are the pixel centres, so the bilinear samples are the pixels themselves
// and the kernel is a streaming blend, 4 pixels (16 B) per lane.
__device__ __forceinline__ uint32_t blend_px(uint32_t a, uint32_t b, float t, float nt)
{
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) o = pack_trunc_u8(nt * ch_f32(a, c) + t * ch_f32(b, c), c, o);
    return o;
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_blend_zero_flow(
    const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out,
    size_t a_stride, size_t b_stride, size_t npx, float t, uint32_t sel)
{
    const float nt = 1.0f - t;
    const uint32_t *pa = reinterpret_cast<const uint32_t *>(a + (size_t)blockIdx.y * a_stride);
    const uint32_t *pb = reinterpret_cast<const uint32_t *>(b + (size_t)blockIdx.y * b_stride);
    uint32_t *po = reinterpret_cast<uint32_t *>(out) + (size_t)blockIdx.y * npx;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= npx) return;
    if (VEC) {
        const uint4 va = *reinterpret_cast<const uint4 *>(pa + i);
        const uint4 vb = *reinterpret_cast<const uint4 *>(pb + i);
        // per-channel arithmetic: swizzling the blended pixel equals blending swizzled inputs
        *reinterpret_cast<uint4 *>(po + i) = swz4(make_uint4(blend_px(va.x, vb.x, t, nt), blend_px(va.y, vb.y, t, nt),
                                                             blend_px(va.z, vb.z, t, nt), blend_px(va.w, vb.w, t, nt)), sel);
    } else {
        po[i] = swz(blend_px(pa[i], pb[i], t, nt), sel);
    }
}

// interpolation/mod.rs:467-510: clamp, bilinear, truncate to u8 -- returned as the four truncated
// channel values still in f32 (floor of a value in [0, 255]) so the blend needs no unpack.
__device__ __forceinline__ float4 sample_trunc(const uint32_t *__restrict__ f, uint32_t w, uint32_t h, float x, float y)
{
    x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    const float xfl = floorf(x), yfl = floorf(y);
    const uint32_t x0 = (uint32_t)xfl, y0 = (uint32_t)yfl;
    const uint32_t x1 = umin(x0 + 1, w - 1), y1 = umin(y0 + 1, h - 1);
    const float xf = x - xfl, yf = y - yfl;
    const float nxf = 1.0f - xf, nyf = 1.0f - yf;
    const uint32_t p00 = f[(size_t)y0 * w + x0], p01 = f[(size_t)y0 * w + x1];
    const uint32_t p10 = f[(size_t)y1 * w + x0], p11 = f[(size_t)y1 * w + x1];
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float top = ch_f32(p00, c) * nxf + ch_f32(p01, c) * xf;
        const float bottom = ch_f32(p10, c) * nxf + ch_f32(p11, c) * xf;
        const float value = top * nyf + bottom * yf;
        r[c] = fminf(floorf(value), 255.0f); // `value as u8`; value >= 0 here
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// Dense flow (2 x f32 per pixel, delta A -> B): A sampled at p - t*flow, B at
// p + (1-t)*flow (warp_blend.wgsl:36-37 in texel space).  blockDim = (64, 4).
__global__ __launch_bounds__(256) void k_warp_blend_flow(
    const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, const float *__restrict__ flow,
    uint8_t *__restrict__ out, size_t a_stride, size_t b_stride, uint32_t w, uint32_t h, float t, uint32_t sel)
{
    const uint32_t y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    const uint32_t x = blockIdx.x * kWave + threadIdx.x;
    if (y >= h || x >= w) return;
    const size_t npx = (size_t)w * h;
    const uint32_t *pa = reinterpret_cast<const uint32_t *>(a + (size_t)blockIdx.z * a_stride);
    const uint32_t *pb = reinterpret_cast<const uint32_t *>(b + (size_t)blockIdx.z * b_stride);
    const size_t idx = (size_t)y * w + x;
    const float2 f = *reinterpret_cast<const float2 *>(flow + ((size_t)blockIdx.z * npx + idx) * 2);
    float tv = t; // per-lane copy: scalar operands halve the VALU issue rate on gfx950
    asm volatile("" : "+v"(tv));
    const float nt = 1.0f - tv;
    const float ax = (float)x - tv * f.x, ay = (float)y - tv * f.y;
    const float bx = (float)x + nt * f.x, by = (float)y + nt * f.y;
    const float4 sa = sample_trunc(pa, w, h, ax, ay);
    const float4 sb = sample_trunc(pb, w, h, bx, by);
    uint32_t o = 0;
    o = pack_trunc_u8(nt * sa.x + tv * sb.x, 0, o);
    o = pack_trunc_u8(nt * sa.y + tv * sb.y, 1, o);
    o = pack_trunc_u8(nt * sa.z + tv * sb.z, 2, o);
    o = pack_trunc_u8(nt * sa.w + tv * sb.w, 3, o);
    reinterpret_cast<uint32_t *>(out)[(size_t)blockIdx.z * npx + idx] = swz(o, sel);
}

// ---------------------------------------------------------------------------------
// BGRA -> RGBA swizzle of captured frames (nu_scaler_core/src/lib.rs:251-270), 4 px per lane
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t swap_rb(uint32_t p)
{
    return __builtin_amdgcn_perm(p, p, 0x03000102u); // bytes (2, 1, 0, 3): one v_perm_b32
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_swizzle_bgra(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t npx)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= npx) return;
    if (VEC) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
        *reinterpret_cast<uint4 *>(out + i) = make_uint4(swap_rb(v.x), swap_rb(v.y), swap_rb(v.z), swap_rb(v.w));
    } else {
        out[i] = swap_rb(in[i]);
    }
}

// ---------------------------------------------------------------------------------
// Optical-flow front end: Gaussian pyramid + Horn-Schunck
// ---------------------------------------------------------------------------------
// Images: f32 RGBA, one float4 (16 B) per pixel per lane; flows: float2 per pixel.
// Straight per-pixel kernels with the shaders' exact expression order (no contraction).

__device__ __forceinline__ float div_by_recip(float x, float y, float z);

// u8 -> f32 / 255 (the Rgba8Unorm view of a frame); exact IEEE quotient via the reciprocal + 2 FMAs
// (equal to x / 255.0f for all 256 inputs, checked on the CPU).
__device__ __forceinline__ float4 unorm8(uint32_t p)
{
    const float z = 1.0f / 255.0f;
    return make_float4(div_by_recip(ch_f32(p, 0), 255.0f, z), div_by_recip(ch_f32(p, 1), 255.0f, z),
                       div_by_recip(ch_f32(p, 2), 255.0f, z), div_by_recip(ch_f32(p, 3), 255.0f, z));
}

__global__ __launch_bounds__(256) void k_rgba8_to_f32(const uint32_t *__restrict__ in, float4 *__restrict__ out, size_t npx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const uint32_t p = in[i];
    out[i] = unorm8(p);
}

__device__ __forceinline__ float4 blur5(const float4 m2, const float4 m1, const float4 c0, const float4 p1, const float4 p2)
{
    // gaussian_blur_h.wgsl:42-46: m2*W0 + m1*W1 + c*W2 + p1*W1 + p2*W0, left to right
    const float W0 = 1.0f / 16.0f, W1 = 4.0f / 16.0f, W2 = 6.0f / 16.0f;
    float4 r;
    r.x = m2.x * W0 + m1.x * W1 + c0.x * W2 + p1.x * W1 + p2.x * W0;
    r.y = m2.y * W0 + m1.y * W1 + c0.y * W2 + p1.y * W1 + p2.y * W0;
    r.z = m2.z * W0 + m1.z * W1 + c0.z * W2 + p1.z * W1 + p2.z * W0;
    r.w = m2.w * W0 + m1.w * W1 + c0.w * W2 + p1.w * W1 + p2.w * W0;
    return r;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// blockDim = (64, 4)
template <bool HORIZONTAL>
__global__ __launch_bounds__(256) void k_blur(const float4 *__restrict__ in, float4 *__restrict__ out, int w, int h)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    float4 t[5];
#pragma unroll
    for (int k = -2; k <= 2; ++k) {
        const int xx = HORIZONTAL ? clampi(x + k, 0, w - 1) : x;
        const int yy = HORIZONTAL ? y : clampi(y + k, 0, h - 1);
        t[k + 2] = in[(size_t)yy * w + xx];
    }
    out[(size_t)y * w + x] = blur5(t[0], t[1], t[2], t[3], t[4]);
}

// downsample.wgsl:22-37, out = ((w+1)/2, (h+1)/2), source clamped at the far edge
__global__ __launch_bounds__(256) void k_downsample(const float4 *__restrict__ in, float4 *__restrict__ out, int w, int h)
{
    const int ow = (w + 1) / 2, oh = (h + 1) / 2;
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const int x0 = 2 * x, y0 = 2 * y, x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const float4 c00 = in[(size_t)y0 * w + x0], c10 = in[(size_t)y0 * w + x1];
    const float4 c01 = in[(size_t)y1 * w + x0], c11 = in[(size_t)y1 * w + x1];
    float4 r;
    r.x = (c00.x + c10.x + c01.x + c11.x) * 0.25f;
    r.y = (c00.y + c10.y + c01.y + c11.y) * 0.25f;
    r.z = (c00.z + c10.z + c01.z + c11.z) * 0.25f;
    r.w = (c00.w + c10.w + c01.w + c11.w) * 0.25f;
    out[(size_t)y * ow + x] = r;
}

__device__ __forceinline__ float lum(const float4 c) { return (c.x + c.y + c.z) * 0.33333f; } // horn_schunck.wgsl:17-20

// One Jacobi step, horn_schunck.wgsl:48-92.
__global__ __launch_bounds__(256) void k_horn_schunck(const float4 *__restrict__ i1, const float4 *__restrict__ i2,
                                                      const float2 *__restrict__ fin, float2 *__restrict__ fout,
                                                      int w, int h, float lambda)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int xp = min(x + 1, w - 1), xm = max(x, 1) - 1, yp = min(y + 1, h - 1), ym = max(y, 1) - 1;
    const float ix = (lum(i1[(size_t)y * w + xp]) - lum(i1[(size_t)y * w + xm])) * 0.5f;
    const float iy = (lum(i1[(size_t)yp * w + x]) - lum(i1[(size_t)ym * w + x])) * 0.5f;
    const float it = lum(i2[(size_t)y * w + x]) - lum(i1[(size_t)y * w + x]);
    float su = 0.0f, sv = 0.0f, count = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const float2 f = fin[(size_t)clampi(y + dy, 0, h - 1) * w + clampi(x + dx, 0, w - 1)];
            su += f.x;
            sv += f.y;
            count += 1.0f;
        }
    const float ua = su / count, va = sv / count;
    const float common = (ix * ua + iy * va + it) / (lambda + ix * ix + iy * iy);
    fout[(size_t)y * w + x] = make_float2(ua - common * ix, va - common * iy);
}

// One pyramid level in one launch (build_pyramid's three dispatches, wgpu_interpolator.rs:1068-1085,
// fused): a 64x16 output tile stages its (64+4)x(16+4) input region in LDS (converted from RGBA8
// at level 0), runs the horizontal blur into a second LDS tile, the vertical blur from there, writes
// the blurred level and, from the same tile, the 2x2-averaged input of the next level.  Each value
// goes through exactly the arithmetic of k_blur<true>, k_blur<false> and k_downsample, so the
// result is bit-identical to the three separate kernels while HBM sees the input once.
constexpr int kPyrTW = 64, kPyrTH = 16;

template <bool U8IN>
__global__ __launch_bounds__(256) void k_pyramid_level(const void *__restrict__ in, float4 *__restrict__ level,
                                                       float4 *__restrict__ next, int w, int h)
{
    __shared__ float4 s_a[(kPyrTH + 4) * (kPyrTW + 4)]; // input region; later the V-blurred tile
    __shared__ float4 s_h[(kPyrTH + 4) * kPyrTW];        // H-blurred rows
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
    const int bx = blockIdx.x * kPyrTW, by = blockIdx.y * kPyrTH;
    // stage input rows by-2 .. by+17, columns bx-2 .. bx+65, coordinates clamped into the image
    for (int r = ty; r < kPyrTH + 4; r += 4) {
        const int gy = clampi(by - 2 + r, 0, h - 1);
        for (int c = tx; c < kPyrTW + 4; c += 64) {
            const int gx = clampi(bx - 2 + c, 0, w - 1);
            const size_t g = (size_t)gy * w + gx;
            s_a[r * (kPyrTW + 4) + c] = U8IN ? unorm8(static_cast<const uint32_t *>(in)[g]) : static_cast<const float4 *>(in)[g];
        }
    }
    __syncthreads();
    // horizontal pass for the 20 staged rows.  The shader clamps x+-k into the image; the staged
    // columns already hold clamp(bx-2+c), so column (tx+2)+k is the clamped neighbour as long as
    // the output column itself is inside the image.
    for (int r = ty; r < kPyrTH + 4; r += 4) {
        const float4 *row = s_a + r * (kPyrTW + 4) + tx;
        s_h[r * kPyrTW + tx] = blur5(row[0], row[1], row[2], row[3], row[4]);
    }
    __syncthreads();
    // vertical pass -> blurred level; keep the tile in LDS (s_a is free now) for the downsample
    const int gx = bx + tx;
    for (int r = ty; r < kPyrTH; r += 4) {
        const float4 v = blur5(s_h[r * kPyrTW + tx], s_h[(r + 1) * kPyrTW + tx], s_h[(r + 2) * kPyrTW + tx],
                               s_h[(r + 3) * kPyrTW + tx], s_h[(r + 4) * kPyrTW + tx]);
        s_a[r * kPyrTW + tx] = v;
        const int gy = by + r;
        if (gx < w && gy < h) level[(size_t)gy * w + gx] = v;
    }
    if (next == nullptr) return; // block-uniform
    __syncthreads();
    // 2x2 box average of the blurred tile (tile origin is even, so every 2x2 block is inside it)
    const int ow = (w + 1) / 2, oh = (h + 1) / 2;
    const int dxl = threadIdx.x & 31, dyl = threadIdx.x >> 5; // 32 x 8 outputs per tile
    const int ox = bx / 2 + dxl, oy = by / 2 + dyl;
    if (ox < ow && oy < oh) {
        const int x0 = 2 * dxl, y0 = 2 * dyl;
        const int x1 = min(bx + x0 + 1, w - 1) - bx, y1 = min(by + y0 + 1, h - 1) - by;
        const float4 c00 = s_a[y0 * kPyrTW + x0], c10 = s_a[y0 * kPyrTW + x1];
        const float4 c01 = s_a[y1 * kPyrTW + x0], c11 = s_a[y1 * kPyrTW + x1];
        float4 r;
        r.x = (c00.x + c10.x + c01.x + c11.x) * 0.25f;
        r.y = (c00.y + c10.y + c01.y + c11.y) * 0.25f;
        r.z = (c00.z + c10.z + c01.z + c11.z) * 0.25f;
        r.w = (c00.w + c10.w + c01.w + c11.w) * 0.25f;
        next[(size_t)oy * ow + ox] = r;
    }
}

// Derivatives of one pyramid level, computed once per level instead of once per Jacobi step:
// (ix, iy, it, lambda + ix*ix + iy*iy) with exactly the expressions of horn_schunck.wgsl:58-82.
__global__ __launch_bounds__(256) void k_hs_prepare(const float4 *__restrict__ i1, const float4 *__restrict__ i2,
                                                    float4 *__restrict__ coef, float *__restrict__ zinv, int w, int h,
                                                    float lambda)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int xp = min(x + 1, w - 1), xm = max(x, 1) - 1, yp = min(y + 1, h - 1), ym = max(y, 1) - 1;
    const float ix = (lum(i1[(size_t)y * w + xp]) - lum(i1[(size_t)y * w + xm])) * 0.5f;
    const float iy = (lum(i1[(size_t)yp * w + x]) - lum(i1[(size_t)ym * w + x])) * 0.5f;
    const float it = lum(i2[(size_t)y * w + x]) - lum(i1[(size_t)y * w + x]);
    const float den = lambda + ix * ix + iy * iy;
    coef[(size_t)y * w + x] = make_float4(ix, iy, it, den);
    zinv[(size_t)y * w + x] = 1.0f / den; // correctly rounded reciprocal, for div_by_recip
}

// K Jacobi steps per launch on an LDS tile (temporal blocking): a 32x32 output tile is loaded
// with a K-cell halo of the current flow and the per-cell coefficients; step j updates the
// cells whose 3x3 neighbourhood was valid after step j-1 (the region shrinks by one ring per
// step, except at the image border where neighbours clamp inwards), ping-ponging between two
// LDS flow buffers.  Same arithmetic and order as k_horn_schunck, so K launches of that
// kernel and one launch of this one produce identical bits.
// Correctly rounded x / y from z = RN(1/y) with one multiply and two FMAs (Markstein): q = RN(x z),
// r = x - q y (exact in the FMA), result = RN(q + r z).  Equal to the IEEE quotient for every finite
// x when y = 9 (checked exhaustively on the CPU) and for every y whose mantissa is not all ones;
// those y take the real division.  Replaces ~11 slow-class instructions per division by 3 fast ones.
__device__ __forceinline__ float div_by_recip(float x, float y, float z)
{
    const float q = x * z;
    const float r = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(r, z, q);
}

struct HsCell {
    float ix, iy, it, den, zinv;
    bool plain_div; // mantissa of den all ones: Markstein's exception
};

// T x T output tile, K Jacobi steps per launch (temporal blocking); blockDim = 256 = 32 x 8 cells
// per sweep.  Each thread owns the same (T+2K)^2 / 256 cells in every step, so their coefficients
// live in registers and only the two ping-pong flow tiles are in LDS.
template <int T, int K>
__global__ __launch_bounds__(256) void k_hs_tiled(const float4 *__restrict__ coef, const float *__restrict__ zinv,
                                                  const float2 *__restrict__ fin, float2 *__restrict__ fout, int w, int h)
{
    constexpr int R = T + 2 * K;
    constexpr int NA = (R + 7) / 8, NB = (R + 31) / 32;
    __shared__ float2 s_flow[2][R * R];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * T - K, y0 = blockIdx.y * T - K; // image coords of LDS cell (0,0)
    HsCell cell[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ly = ty + 8 * a, lx = tx + 32 * b;
            if (ly < R && lx < R) {
                const int gx = clampi(x0 + lx, 0, w - 1), gy = clampi(y0 + ly, 0, h - 1);
                const size_t g = (size_t)gy * w + gx;
                const float4 c = coef[g];
                cell[a][b].ix = c.x;
                cell[a][b].iy = c.y;
                cell[a][b].it = c.z;
                cell[a][b].den = c.w;
                cell[a][b].zinv = zinv[g];
                cell[a][b].plain_div = (__float_as_uint(c.w) & 0x7fffffu) == 0x7fffffu;
                s_flow[0][ly * R + lx] = fin[g];
            }
        }
    __syncthreads();
    // tiles whose loaded region lies strictly inside the image need no clamping at all
    const bool border = x0 < 0 || y0 < 0 || x0 + R > w || y0 + R > h; // block-uniform
    int cur = 0;
#pragma unroll 1
    for (int j = 1; j <= K; ++j) {
        // step j updates cells [j, R-j) of the tile (plus whatever the image border clamps inwards)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int ly = ty + 8 * a, lx = tx + 32 * b;
                if (ly < j || ly >= R - j || lx < j || lx >= R - j) continue;
                const int gx = x0 + lx, gy = y0 + ly;
                if (border && (gx < 0 || gy < 0 || gx >= w || gy >= h)) continue;
                float su = 0.0f, sv = 0.0f;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        int nx = lx + dx, ny = ly + dy;
                        if (border) {
                            nx = clampi(gx + dx, 0, w - 1) - x0;
                            ny = clampi(gy + dy, 0, h - 1) - y0;
                        }
                        const float2 f = s_flow[cur][ny * R + nx];
                        su += f.x;
                        sv += f.y;
                    }
                // sum / count with count == 9 (horn_schunck.wgsl:38-41)
                const float ua = div_by_recip(su, 9.0f, 1.0f / 9.0f), va = div_by_recip(sv, 9.0f, 1.0f / 9.0f);
                const HsCell &c = cell[a][b];
                const float num = c.ix * ua + c.iy * va + c.it;
                const float common = c.plain_div ? num / c.den : div_by_recip(num, c.den, c.zinv);
                s_flow[cur ^ 1][ly * R + lx] = make_float2(ua - common * c.ix, va - common * c.iy);
            }
        __syncthreads();
        cur ^= 1;
    }
    for (int ly = ty + K; ly < T + K; ly += 8)
        for (int lx = tx + K; lx < T + K; lx += 32) {
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < w && gy < h) fout[(size_t)gy * w + gx] = s_flow[cur][ly * R + lx];
        }
}

// flow_upsample.wgsl:27-36 (linear clamp-to-edge sampler in texel space), vectors * scale
__global__ __launch_bounds__(256) void k_flow_upsample(const float2 *__restrict__ src, int sw, int sh,
                                                       float2 *__restrict__ dst, int dw, int dh, float scale)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
    const float sx = u * (float)sw - 0.5f, sy = v * (float)sh - 0.5f;
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const float fx = sx - fx0, fy = sy - fy0;
    const int x0 = clampi((int)fx0, 0, sw - 1), x1 = clampi((int)fx0 + 1, 0, sw - 1);
    const int y0 = clampi((int)fy0, 0, sh - 1), y1 = clampi((int)fy0 + 1, 0, sh - 1);
    const float2 a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1];
    const float2 c = src[(size_t)y1 * sw + x0], d = src[(size_t)y1 * sw + x1];
    float2 r;
    r.x = ((a.x * (1.0f - fx) + b.x * fx) * (1.0f - fy) + (c.x * (1.0f - fx) + d.x * fx) * fy) * scale;
    r.y = ((a.y * (1.0f - fx) + b.y * fx) * (1.0f - fy) + (c.y * (1.0f - fx) + d.y * fx) * fy) * scale;
    dst[(size_t)y * dw + x] = r;
}

constexpr uint32_t kMaxGridZ = 65535;

} // namespace

// ---------------------------------------------------------------------------------
// FSR1-style EASU + RCAS: nu_scaler_core/src/upscale/fsr.rs:24-260
// ---------------------------------------------------------------------------------
// One kernel, three modes.  Stage 1 fills an LDS tile of packed RGBA8 pixels -- EASU evaluations
// (modes Easu, Fused) or plain loads (mode Rcas) -- with a 1-pixel halo when RCAS follows; stage 2
// writes the tile out, through the 5-tap RCAS when asked.  The fused mode therefore never writes the
// EASU image to HBM (the shader pair round-trips it as RGBA8, which the LDS tile reproduces exactly:
// same truncating pack between the passes).  Expression order follows the shaders; no contraction.
namespace {

enum class FsrMode : int { Easu = 0, Rcas = 1, Fused = 2 };

constexpr int kFsrTW = 64, kFsrTH = 32; // output tile per 256-thread block

struct FsrArgs {
    const uint32_t *in;
    uint32_t *out;
    int iw, ih, ow, oh;
    size_t ipx, opx;   // pixels per input / output frame
    float sx, sy;      // f32(iw) / f32(ow), f32(ih) / f32(oh)   (host, IEEE)
    float easu_sharp, rcas_sharp;
    uint32_t sel;      // input channel order (kSelRGBA / kSelBGRA)
};

__device__ __forceinline__ float3 fsr_rgb(uint32_t p)
{
    const float z = 1.0f / 255.0f;
    return make_float3(div_by_recip(ch_f32(p, 0), 255.0f, z), div_by_recip(ch_f32(p, 1), 255.0f, z),
                       div_by_recip(ch_f32(p, 2), 255.0f, z));
}

// fsr.rs:74-84.  The shader's 0.5 * d3 and 0.0 - 0.5 * d are exact (scaling by a power of two), so
// folding each into the neighbouring add as an FMA rounds exactly as the two separate operations do.
__device__ __forceinline__ float fsr_cubic(float d)
{
    const float d2 = d * d;
    const float d3 = d * d2;
    const float near = __builtin_fmaf(-0.5f, d3, 2.0f - 1.5f * d) + d2; // 2.0 - 1.5*d - 0.5*d3 + d2
    const float far = __builtin_fmaf(-0.5f, d, 2.5f * d2) - d3;         // 0.0 - 0.5*d + 2.5*d2 - d3
    return d <= 1.0f ? near : (d <= 2.0f ? far : 0.0f);
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// pack_rgba8(vec4(rgb, 1.0)): u32(clamp(v, 0, 1) * 255.0) per channel
__device__ __forceinline__ uint32_t fsr_pack(float r, float g, float b)
{
    uint32_t o = 0xff000000u;
    o = pack_trunc_u8(clamp01(r) * 255.0f, 0, o);
    o = pack_trunc_u8(clamp01(g) * 255.0f, 1, o);
    o = pack_trunc_u8(clamp01(b) * 255.0f, 2, o);
    return o;
}

// EASU direction weight wx of FsrDirA + fsr.rs:131-133, from the four unpacked neighbours
__device__ __forceinline__ float fsr_dir_wx(const float3 up, const float3 dn, const float3 lf, const float3 rt)
{
    const float vgx = (fabsf(up.x - dn.x) + fabsf(up.y - dn.y) + fabsf(up.z - dn.z)) / 3.0f;
    const float vgy = (fabsf(lf.x - rt.x) + fabsf(lf.y - rt.y) + fabsf(lf.z - rt.z)) / 3.0f;
    const float dxr = vgx + 0.0001f, dyr = vgy + 0.0001f;
    const float len = sqrtf(dxr * dxr + dyr * dyr);
    const float dirx = dxr / len, diry = dyr / len;
    return fabsf(dirx) / (fabsf(dirx) + fabsf(diry));
}

// x / den for three numerators: one correctly rounded reciprocal, then Markstein's correction per
// numerator (div_by_recip); a den whose mantissa is all ones takes the plain division.
__device__ __forceinline__ void fsr_div3(float &r, float &g, float &b, float den)
{
    if ((__float_as_uint(den) & 0x007fffffu) == 0x007fffffu) {
        r = r / den;
        g = g / den;
        b = b / den;
    } else {
        const float z = 1.0f / den;
        r = div_by_recip(r, den, z);
        g = div_by_recip(g, den, z);
        b = div_by_recip(b, den, z);
    }
}

// The 16-tap sum of fsr.rs:135-161 given the taps (float4: rgb in xyz) through `tap(x, y)`.
template <typename TAP>
__device__ __forceinline__ uint32_t fsr_easu_taps(TAP &&tap, float wx, float fx, float fy, float sharp)
{
    const float wy = 1.0f - wx;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const float pyw = ((float)y - fy) * wy;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float4 c = tap(x, y);
            const float dist = fabsf(((float)x - fx) * wx + pyw);
            const float wgt = fsr_cubic(dist);
            sr = sr + c.x * wgt;
            sg = sg + c.y * wgt;
            sb = sb + c.z * wgt;
            sw = sw + wgt;
        }
    }
    fsr_div3(sr, sg, sb, fmaxf(sw, 0.0001f));
    if (sharp > 0.001f) {
        const float4 ctr = tap(1, 1);
        const float ns = 1.0f - sharp;
        sr = sr * ns + ctr.x * sharp;
        sg = sg * ns + ctr.y * sharp;
        sb = sb * ns + ctr.z * sharp;
    }
    return fsr_pack(sr, sg, sb);
}

// EASU at output pixel (gx, gy), taps straight from memory: fsr.rs:104-169
__device__ __forceinline__ uint32_t fsr_easu_px(const uint32_t *__restrict__ in, const FsrArgs &A, int gx, int gy)
{
    const float cx = ((float)gx + 0.5f) * A.sx, cy = ((float)gy + 0.5f) * A.sy;
    const int ix = (int)cx, iy = (int)cy;
    const float fx = cx - floorf(cx), fy = cy - floorf(cy);
    int xs[4], ys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xs[k] = clampi(ix - 1 + k, 0, A.iw - 1);
        ys[k] = clampi(iy - 1 + k, 0, A.ih - 1);
    }
    float4 t[4][4];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float3 c = fsr_rgb(swz(in[(size_t)ys[y] * A.iw + xs[x]], A.sel));
            t[y][x] = make_float4(c.x, c.y, c.z, 0.0f);
        }
    // FsrDirA at (ix, iy): its four neighbours are taps (1,0) (1,2) (0,1) (2,1) of the 4x4 block
    // (ix, iy always lie inside the image)
    auto rgb = [](const float4 v) { return make_float3(v.x, v.y, v.z); };
    const float wx = fsr_dir_wx(rgb(t[0][1]), rgb(t[2][1]), rgb(t[1][0]), rgb(t[1][2]));
    return fsr_easu_taps([&](int x, int y) { return t[y][x]; }, wx, fx, fy, A.easu_sharp);
}

__device__ __forceinline__ float fsr_luma(const float3 c) { return c.x * 0.299f + c.y * 0.587f + c.z * 0.114f; }

// RCAS from its five taps, already unpacked (rgb in xyz, luma in w): fsr.rs:218-260
__device__ __forceinline__ float4 fsr_rcas_tap(uint32_t p)
{
    const float3 c = fsr_rgb(p);
    return make_float4(c.x, c.y, c.z, fsr_luma(c));
}

__device__ __forceinline__ uint32_t fsr_rcas_px(const float4 c, const float4 t, const float4 b, const float4 l, const float4 r,
                                                float sharp)
{
    const float mn = fminf(c.w, fminf(fminf(t.w, b.w), fminf(l.w, r.w)));
    const float mx = fmaxf(c.w, fmaxf(fmaxf(t.w, b.w), fmaxf(l.w, r.w)));
    // (contrast - 0.0) / (0.2 - 0.0): x - 0.0 == x; the quotient by the constant via its reciprocal (exact, see div_by_recip)
    const float st = clamp01(div_by_recip(mx - mn, 0.2f, 1.0f / 0.2f));
    const float strength = sharp * (1.0f - st * st * (3.0f - 2.0f * st));
    return fsr_pack(c.x + (4.0f * c.x - t.x - b.x - l.x - r.x) * strength,
                    c.y + (4.0f * c.y - t.y - b.y - l.y - r.y) * strength,
                    c.z + (4.0f * c.z - t.z - b.z - l.z - r.z) * strength);
}

// SRC_LDS: the input footprint of the tile is unpacked once into LDS as float4 (rgb / 255, and in .w
// the direction weight wx of that texel), so an EASU evaluation is 16 ds_read_b128 + the tap sum; the
// host picks it when the footprint of every tile fits `src_cap` texels (up-scaling; see launch_fsr1).
template <FsrMode MODE, bool VEC, bool SRC_LDS>
__global__ __launch_bounds__(256) void k_fsr1(const FsrArgs A, const int src_cap)
{
    constexpr int HALO = MODE == FsrMode::Easu ? 0 : 1;
    constexpr int LW = kFsrTW + 2 * HALO, LH = kFsrTH + 2 * HALO;
    // Easu: packed pixels; Rcas / Fused: the pass-1 pixel re-unpacked once (what RCAS reads) + its luma
    using TilePx = typename std::conditional<MODE == FsrMode::Easu, uint32_t, float4>::type;
    __shared__ TilePx tile[LW * LH];
    extern __shared__ float4 src[]; // SRC_LDS: src_cap texels
    const uint32_t *__restrict__ in = A.in + (size_t)blockIdx.z * A.ipx;
    uint32_t *__restrict__ out = A.out + (size_t)blockIdx.z * A.opx;
    const int x0 = blockIdx.x * kFsrTW, y0 = blockIdx.y * kFsrTH;
    const int tid = threadIdx.x;
    int fx0 = 0, fy0 = 0, fw = 0;
    if (MODE != FsrMode::Rcas && SRC_LDS) {
        // footprint of the (clamped) output range of this tile: taps ix-1 .. ix+2 of its first / last pixel
        const int gx_lo = clampi(x0 - HALO, 0, A.ow - 1), gx_hi = clampi(x0 + kFsrTW - 1 + HALO, 0, A.ow - 1);
        const int gy_lo = clampi(y0 - HALO, 0, A.oh - 1), gy_hi = clampi(y0 + kFsrTH - 1 + HALO, 0, A.oh - 1);
        fx0 = (int)(((float)gx_lo + 0.5f) * A.sx) - 1;
        fy0 = (int)(((float)gy_lo + 0.5f) * A.sy) - 1;
        fw = (int)(((float)gx_hi + 0.5f) * A.sx) + 2 - fx0 + 1;
        const int fh = (int)(((float)gy_hi + 0.5f) * A.sy) + 2 - fy0 + 1;
        const int n = fw * fh; // <= src_cap (host-checked)
        for (int i = tid; i < n; i += 256) {
            const int ly = i / fw, lx = i - ly * fw;
            const float3 c = fsr_rgb(swz(in[(size_t)clampi(fy0 + ly, 0, A.ih - 1) * A.iw + clampi(fx0 + lx, 0, A.iw - 1)], A.sel));
            src[i] = make_float4(c.x, c.y, c.z, 0.0f);
        }
        __syncthreads();
        // direction weight of every texel that can be a tile pixel's (ix, iy): the footprint's interior
        for (int i = tid; i < n; i += 256) {
            const int ly = i / fw, lx = i - ly * fw;
            if (lx < 1 || ly < 1 || lx > fw - 2 || ly > fh - 2) continue;
            auto rgb = [&](int j) { const float4 v = src[j]; return make_float3(v.x, v.y, v.z); };
            src[i].w = fsr_dir_wx(rgb(i - fw), rgb(i + fw), rgb(i - 1), rgb(i + 1));
        }
        __syncthreads();
    }
    (void)src_cap;
    // stage 1: the tile (+ halo), coordinates clamped into the image as both shaders' fetches do
    for (int i = tid; i < LW * LH; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = clampi(x0 + lx - HALO, 0, A.ow - 1), gy = clampi(y0 + ly - HALO, 0, A.oh - 1);
        uint32_t p;
        if (MODE == FsrMode::Rcas) {
            p = swz(in[(size_t)gy * A.ow + gx], A.sel);
        } else if (SRC_LDS) {
            const float cx = ((float)gx + 0.5f) * A.sx, cy = ((float)gy + 0.5f) * A.sy;
            const int ix = (int)cx, iy = (int)cy;
            const float4 *t = src + (iy - 1 - fy0) * fw + (ix - 1 - fx0);
            p = fsr_easu_taps([&](int x, int y) { return t[y * fw + x]; }, t[fw + 1].w, cx - floorf(cx),
                              cy - floorf(cy), A.easu_sharp);
        } else {
            p = fsr_easu_px(in, A, gx, gy);
        }
        if constexpr (MODE == FsrMode::Easu)
            tile[i] = p;
        else
            tile[i] = fsr_rcas_tap(p);
    }
    __syncthreads();
    // stage 2: 4 pixels per thread, 16 threads per row, 16 rows per sweep
    const int qx = (tid & 15) * 4, qy = tid >> 4;
#pragma unroll
    for (int sweep = 0; sweep < kFsrTH / 16; ++sweep) {
        const int ly = qy + sweep * 16, gy = y0 + ly;
        if (gy >= A.oh || x0 + qx >= A.ow) continue;
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (ly + HALO) * LW + qx + k + HALO;
            if constexpr (MODE == FsrMode::Easu)
                px[k] = tile[c];
            else
                px[k] = fsr_rcas_px(tile[c], tile[c - LW], tile[c + LW], tile[c - 1], tile[c + 1], A.rcas_sharp);
        }
        uint32_t *dst = out + (size_t)gy * A.ow + x0 + qx;
        if (VEC) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(px[0], px[1], px[2], px[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x0 + qx + k < A.ow) dst[k] = px[k];
        }
    }
}

} // namespace

const char *variant_name(Variant v)
{
    switch (v) {
    case Variant::NearestTable: return "nearest_table";
    case Variant::NearestX2: return "nearest_x2_vec16";
    case Variant::BilinearTable: return "bilinear_table_f32";
    case Variant::BilinearX2Int: return "bilinear_x2_packed_u8";
    case Variant::LanczosGeneral: return "lanczos3_general";
    case Variant::ResizeRows: return "resize_rows_lds";
    case Variant::LanczosX2RegWin: return "lanczos3_x2_regwin";
    case Variant::FsrEasu: return "fsr1_easu_tile";
    case Variant::FsrRcas: return "fsr1_rcas_tile";
    case Variant::Fsr1Fused: return "fsr1_easu_rcas_fused_lds";
    }
    return "?";
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Frames go on grid.z (<= 65535 per launch); longer batches are issued in chunks.
static thread_local uint32_t g_chunk_first_frame = 0; // index of the chunk's first frame, for second inputs
template <typename F>
static hipError_t for_frame_chunks(const UpscaleLaunch &L, F &&f)
{
    const size_t in_bytes = L.in_stride ? L.in_stride : (size_t)L.iw * L.ih * 4, out_bytes = (size_t)L.ow * L.oh * 4;
    for (uint32_t done = 0; done < L.n_frames;) {
        const uint32_t n = L.n_frames - done < kMaxGridZ ? L.n_frames - done : kMaxGridZ;
        g_chunk_first_frame = done;
        f(L.in + (size_t)done * in_bytes, L.out + (size_t)done * out_bytes, n);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += n;
    }
    return hipSuccess;
}

// Output rows per wave of the row-walking table kernels: tall enough that the lerped / gathered source
// rows get reused, small enough that the launch still has a few thousand waves.
static uint32_t rows_per_wave_for(const UpscaleLaunch &L, uint32_t cols_per_wave, uint32_t n_frames)
{
    const uint64_t strips = cdiv(L.ow, cols_per_wave);
    const uint64_t t = (uint64_t)L.oh * strips * n_frames / 8192;
    return (uint32_t)(t < 8 ? 8 : (t > 64 ? 64 : t));
}

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

hipError_t launch_nearest_x2(const UpscaleLaunch &L)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const dim3 block(kWave, 4), grid(cdiv(L.iw, 256), cdiv(L.ih, 4), n);
        hipLaunchKernelGGL(k_nearest_x2, grid, block, 0, L.stream, reinterpret_cast<const uint32_t *>(in),
                           reinterpret_cast<uint32_t *>(out), L.iw, L.ih, ipx, opx, L.in_sel);
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
#define NUS_BL(V, W)                                                                                         \
    hipLaunchKernelGGL((k_bilinear_table<V, W>), grid, block, 0, L.stream, i32, o32, T.bl_x0, T.bl_fx, T.bl_y0, \
                       T.bl_fy, L.iw, L.ih, L.ow, L.oh, rpw, ipx, opx, L.in_sel)
        if (vec && wgsl_form) NUS_BL(true, true);
        else if (vec) NUS_BL(true, false);
        else if (wgsl_form) NUS_BL(false, true);
        else NUS_BL(false, false);
#undef NUS_BL
    });
}

hipError_t launch_bilinear_x2_int(const UpscaleLaunch &L)
{
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const dim3 block(kWave, 4), grid(cdiv(L.iw, 256), cdiv(L.ih, 4), n);
        hipLaunchKernelGGL(k_bilinear_x2_int, grid, block, 0, L.stream, reinterpret_cast<const uint32_t *>(in),
                           reinterpret_cast<uint32_t *>(out), L.iw, L.ih, ipx, opx, L.in_sel);
    });
}

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

hipError_t launch_resize_rows(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max, bool small_taps)
{
    const bool vec = (L.ow % 4) == 0;
    const size_t ipx = (size_t)L.iw * L.ih, opx = (size_t)L.ow * L.oh;
    const uint32_t segw = vec ? 256 : 64;
    const size_t lds = (size_t)4 * (ncols_max + 8) * sizeof(float4);
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        const uint64_t blocks_x = cdiv(cdiv(L.ow, segw), 4);
        uint64_t rpb = (uint64_t)L.oh * blocks_x * n / 4096; // a few thousand blocks per launch
        rpb = rpb < 4 ? 4 : (rpb > 32 ? 32 : rpb);
        const dim3 block(kWave, 4), grid((uint32_t)blocks_x, cdiv(L.oh, (uint32_t)rpb), n);
        auto *i32 = reinterpret_cast<const uint32_t *>(in);
        auto *o32 = reinterpret_cast<uint32_t *>(out);
#define NUS_RR(E, V, S)                                                                                             \
    hipLaunchKernelGGL((k_resize_rows<E, V, S>), grid, block, lds, L.stream, i32, o32, T.lz_lx, T.lz_nx, T.lz_wx, T.lz_ly, \
                       T.lz_ny, T.lz_wy, T.lz_stride, L.iw, L.ow, L.oh, (uint32_t)rpb, ncols_max, ipx, opx, L.in_sel)
        if (exact) {
            if (vec) { if (small_taps) NUS_RR(true, true, true); else NUS_RR(true, true, false); }
            else { if (small_taps) NUS_RR(true, false, true); else NUS_RR(true, false, false); }
        } else {
            if (vec) { if (small_taps) NUS_RR(false, true, true); else NUS_RR(false, true, false); }
            else { if (small_taps) NUS_RR(false, false, true); else NUS_RR(false, false, false); }
        }
#undef NUS_RR
    });
}

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
    // dev knob: unused dynamic LDS per block, to study occupancy sensitivity (0 in production)
    static const uint32_t lds_pad = getenv("NUS_LDS_PAD_KB") ? (uint32_t)atoi(getenv("NUS_LDS_PAD_KB")) * 1024u : 0u;
    hipError_t e = for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = in;
        A.in_b = L.in_b ? L.in_b + (size_t)g_chunk_first_frame * L.in_b_stride : nullptr;
        A.out = out;
        const dim3 block(256), grid(cdiv(nwaves, 4), n);
#define NUS_LZ(E, B) hipLaunchKernelGGL((k_lanczos3_x2<E, B>), grid, block, lds_pad, L.stream, A)
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
        A.in_b = L.in_b ? L.in_b + (size_t)g_chunk_first_frame * L.in_b_stride : nullptr;
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

// Largest input footprint (texels) of any kFsrTW x kFsrTH tile (+halo), with the kernel's own f32 index math.
static size_t fsr_max_footprint(const FsrArgs &A, int halo)
{
    auto span = [&](int out_n, int tile, float scale) {
        int worst = 0;
        for (int o0 = 0; o0 < out_n; o0 += tile) {
            const int lo = o0 - halo < 0 ? 0 : o0 - halo;
            const int hi = o0 + tile - 1 + halo > out_n - 1 ? out_n - 1 : o0 + tile - 1 + halo;
            const int n = (int)(((float)hi + 0.5f) * scale) + 2 - ((int)(((float)lo + 0.5f) * scale) - 1) + 1;
            if (n > worst) worst = n;
        }
        return worst;
    };
    return (size_t)span(A.ow, kFsrTW, A.sx) * (size_t)span(A.oh, kFsrTH, A.sy);
}

hipError_t launch_fsr1(const UpscaleLaunch &L, int mode, float easu_sharpness, float rcas_sharpness)
{
    FsrArgs A;
    A.iw = (int)L.iw;
    A.ih = (int)L.ih;
    A.ow = (int)L.ow;
    A.oh = (int)L.oh;
    A.ipx = (size_t)L.iw * L.ih;
    A.opx = (size_t)L.ow * L.oh;
    A.sx = (float)L.iw / (float)L.ow;
    A.sy = (float)L.ih / (float)L.oh;
    A.easu_sharp = easu_sharpness;
    A.rcas_sharp = rcas_sharpness;
    A.sel = L.in_sel;
    const bool vec = (L.ow % 4) == 0;
    // LDS source tile when every tile's footprint fits beside the pixel tile in 64 KiB: 3072 texels for
    // EASU alone (any up-scaling ratio), 1700 next to the fused mode's float4 tile (ratios >= ~1.3)
    const size_t foot = mode == 1 ? 0 : fsr_max_footprint(A, mode == 2 ? 1 : 0);
    const bool src_lds = mode != 1 && foot <= (mode == 2 ? 1700u : 3072u);
    const size_t dyn = src_lds ? foot * sizeof(float4) : 0;
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = reinterpret_cast<const uint32_t *>(in);
        A.out = reinterpret_cast<uint32_t *>(out);
        const dim3 block(256), grid(cdiv(L.ow, kFsrTW), cdiv(L.oh, kFsrTH), n);
#define NUS_FSR2(M, V, S) hipLaunchKernelGGL((k_fsr1<M, V, S>), grid, block, dyn, L.stream, A, (int)foot)
#define NUS_FSR(M)                      \
    if (vec && src_lds)                 \
        NUS_FSR2(M, true, true);        \
    else if (vec)                       \
        NUS_FSR2(M, true, false);       \
    else if (src_lds)                   \
        NUS_FSR2(M, false, true);       \
    else                                \
        NUS_FSR2(M, false, false)
        if (mode == 0) {
            NUS_FSR(FsrMode::Easu);
        } else if (mode == 1) {
            if (vec)
                NUS_FSR2(FsrMode::Rcas, true, false);
            else
                NUS_FSR2(FsrMode::Rcas, false, false);
        } else {
            NUS_FSR(FsrMode::Fused);
        }
#undef NUS_FSR
#undef NUS_FSR2
    });
}

hipError_t launch_swizzle_bgra(const uint8_t *in, uint8_t *out, size_t npx, hipStream_t stream)
{
    const bool vec = (npx % 4) == 0 && (reinterpret_cast<uintptr_t>(in) % 16) == 0 && (reinterpret_cast<uintptr_t>(out) % 16) == 0;
    const size_t items = vec ? npx / 4 : npx;
    const dim3 block(256), grid((uint32_t)((items + 255) / 256));
    if (vec)
        hipLaunchKernelGGL(k_swizzle_bgra<true>, grid, block, 0, stream, reinterpret_cast<const uint32_t *>(in), reinterpret_cast<uint32_t *>(out), npx);
    else
        hipLaunchKernelGGL(k_swizzle_bgra<false>, grid, block, 0, stream, reinterpret_cast<const uint32_t *>(in), reinterpret_cast<uint32_t *>(out), npx);
    return hipGetLastError();
}

hipError_t launch_rgba8_to_f32(const uint8_t *in, float *out, uint32_t w, uint32_t h, hipStream_t stream)
{
    const size_t npx = (size_t)w * h;
    hipLaunchKernelGGL(k_rgba8_to_f32, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(in), reinterpret_cast<float4 *>(out), npx);
    return hipGetLastError();
}

hipError_t launch_blur(const float *in, float *out, uint32_t w, uint32_t h, bool horizontal, hipStream_t stream)
{
    const dim3 block(kWave, 4), grid(cdiv(w, kWave), cdiv(h, 4));
    if (horizontal)
        hipLaunchKernelGGL(k_blur<true>, grid, block, 0, stream, reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), (int)w, (int)h);
    else
        hipLaunchKernelGGL(k_blur<false>, grid, block, 0, stream, reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), (int)w, (int)h);
    return hipGetLastError();
}

hipError_t launch_downsample(const float *in, float *out, uint32_t w, uint32_t h, hipStream_t stream)
{
    const dim3 block(kWave, 4), grid(cdiv((w + 1) / 2, kWave), cdiv((h + 1) / 2, 4));
    hipLaunchKernelGGL(k_downsample, grid, block, 0, stream, reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), (int)w, (int)h);
    return hipGetLastError();
}

hipError_t launch_horn_schunck(const float *i1, const float *i2, const float *flow_in, float *flow_out, uint32_t w,
                               uint32_t h, float lambda, hipStream_t stream)
{
    const dim3 block(kWave, 4), grid(cdiv(w, kWave), cdiv(h, 4));
    hipLaunchKernelGGL(k_horn_schunck, grid, block, 0, stream, reinterpret_cast<const float4 *>(i1),
                       reinterpret_cast<const float4 *>(i2), reinterpret_cast<const float2 *>(flow_in),
                       reinterpret_cast<float2 *>(flow_out), (int)w, (int)h, lambda);
    return hipGetLastError();
}

// One fused pyramid level: `in` is RGBA8 (u8_input) or f32 RGBA; `next` may be null (last level).
hipError_t launch_pyramid_level(const void *in, bool u8_input, float *level, float *next, uint32_t w, uint32_t h,
                                hipStream_t stream)
{
    const dim3 block(256), grid(cdiv(w, kPyrTW), cdiv(h, kPyrTH));
    if (u8_input)
        hipLaunchKernelGGL(k_pyramid_level<true>, grid, block, 0, stream, in, reinterpret_cast<float4 *>(level),
                           reinterpret_cast<float4 *>(next), (int)w, (int)h);
    else
        hipLaunchKernelGGL(k_pyramid_level<false>, grid, block, 0, stream, in, reinterpret_cast<float4 *>(level),
                           reinterpret_cast<float4 *>(next), (int)w, (int)h);
    return hipGetLastError();
}

// coef: w*h float4 followed by w*h floats (reciprocals) -> w*h*20 bytes
hipError_t launch_hs_prepare(const float *i1, const float *i2, float *coef, uint32_t w, uint32_t h, float lambda,
                             hipStream_t stream)
{
    const dim3 block(kWave, 4), grid(cdiv(w, kWave), cdiv(h, 4));
    hipLaunchKernelGGL(k_hs_prepare, grid, block, 0, stream, reinterpret_cast<const float4 *>(i1),
                       reinterpret_cast<const float4 *>(i2), reinterpret_cast<float4 *>(coef),
                       coef + (size_t)w * h * 4, (int)w, (int)h, lambda);
    return hipGetLastError();
}

// `iterations` Jacobi steps from *flow_a, ping-ponging with *flow_b; on return *flow_a holds the
// result (the pointers are swapped as needed).  Steps are grouped 8 / 4 / 2 / 1 per launch; small
// levels use 16x16 tiles so that the grid still covers the 256 CUs.
hipError_t launch_hs_iterate(const float *coef, float **flow_a, float **flow_b, uint32_t w, uint32_t h,
                             uint32_t iterations, hipStream_t stream)
{
    const bool small = (uint64_t)cdiv(w, 32) * cdiv(h, 32) < 1024;
    const uint32_t T = small ? 16 : 32;
    const dim3 block(256), grid(cdiv(w, T), cdiv(h, T));
    auto c4 = reinterpret_cast<const float4 *>(coef);
    const float *zi = coef + (size_t)w * h * 4;
    uint32_t launches = (iterations + 7) / 8;
    while (iterations > 0) {
        auto fi = reinterpret_cast<const float2 *>(*flow_a);
        auto fo = reinterpret_cast<float2 *>(*flow_b);
        const uint32_t k = (iterations + launches - 1) / launches; // even split, 1..8 steps per launch
#define NUS_HS(KK)                                                                                              \
    case KK:                                                                                                    \
        if (small)                                                                                              \
            hipLaunchKernelGGL((k_hs_tiled<16, KK>), grid, block, 0, stream, c4, zi, fi, fo, (int)w, (int)h);   \
        else                                                                                                    \
            hipLaunchKernelGGL((k_hs_tiled<32, KK>), grid, block, 0, stream, c4, zi, fi, fo, (int)w, (int)h);   \
        break;
        switch (k) {
            NUS_HS(1) NUS_HS(2) NUS_HS(3) NUS_HS(4) NUS_HS(5) NUS_HS(6) NUS_HS(7) NUS_HS(8)
        }
#undef NUS_HS
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        iterations -= k;
        --launches;
        float *t = *flow_a;
        *flow_a = *flow_b;
        *flow_b = t;
    }
    return hipSuccess;
}

hipError_t launch_flow_upsample(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh,
                                float scale, hipStream_t stream)
{
    const dim3 block(kWave, 4), grid(cdiv(dw, kWave), cdiv(dh, 4));
    hipLaunchKernelGGL(k_flow_upsample, grid, block, 0, stream, reinterpret_cast<const float2 *>(src), (int)sw, (int)sh,
                       reinterpret_cast<float2 *>(dst), (int)dw, (int)dh, scale);
    return hipGetLastError();
}

hipError_t launch_warp_blend(const WarpLaunch &L)
{
    const size_t npx = (size_t)L.w * L.h;
    for (uint32_t done = 0; done < L.n_pairs;) {
        const uint32_t n = L.n_pairs - done < kMaxGridZ ? L.n_pairs - done : kMaxGridZ;
        const uint8_t *a = L.a + (size_t)done * L.a_stride;
        const uint8_t *b = L.b + (size_t)done * L.b_stride;
        uint8_t *out = L.out + (size_t)done * npx * 4;
        if (L.flow == nullptr) {
            const bool vec = (npx % 4) == 0 && (L.a_stride % 16) == 0 && (L.b_stride % 16) == 0 &&
                             (reinterpret_cast<uintptr_t>(a) % 16) == 0 && (reinterpret_cast<uintptr_t>(b) % 16) == 0 &&
                             (reinterpret_cast<uintptr_t>(out) % 16) == 0;
            const size_t items = vec ? npx / 4 : npx;
            const dim3 block(256), grid((uint32_t)((items + 255) / 256), n);
            if (vec)
                hipLaunchKernelGGL(k_blend_zero_flow<true>, grid, block, 0, L.stream, a, b, out, L.a_stride, L.b_stride, npx, L.t, L.in_sel);
            else
                hipLaunchKernelGGL(k_blend_zero_flow<false>, grid, block, 0, L.stream, a, b, out, L.a_stride, L.b_stride, npx, L.t, L.in_sel);
        } else {
            const dim3 block(kWave, 4), grid(cdiv(L.w, 64), cdiv(L.h, 4), n);
            hipLaunchKernelGGL(k_warp_blend_flow, grid, block, 0, L.stream, a, b, L.flow + (size_t)done * npx * 2, out,
                               L.a_stride, L.b_stride, L.w, L.h, L.t, L.in_sel);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += n;
    }
    return hipSuccess;
}

} // namespace nus

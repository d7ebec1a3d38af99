// nus_warp_device.hpp -- the dense-flow warp + blend of ONE pixel (device code shared by k_warp_blend_flow, nus_k_interp.hip, and by
// the Jacobi kernel that warps with the flow it has just finished, nus_k_flow.hip): the same expressions in both, so the same bytes.
//   geometry nu_scaler_core/src/shaders/warp_blend.wgsl:25-43, rounding nu_scaler_core/src/interpolation/mod.rs:386-411, :467-510
#pragma once

#include "nus_device.hpp"

namespace nus {

namespace {

constexpr int kWarpExact = 0, kWarpFma = 1;

template <int MODE>
__device__ __forceinline__ float lerp_mode(float a, float b, float f, float nf)
{
    if (MODE == kWarpFma) return __builtin_fmaf(b, f, a * nf); // (the form a + f (b - a) makes the compiler subtract the packed
                                                               // bytes and convert the difference: two slow-class instructions)
    return a * nf + b * f;
}

template <int MODE>
__device__ __forceinline__ float4 sample_corner(__amdgpu_buffer_rsrc_t rs, uint32_t row_bytes, float wmax, float hmax,
                                                uint32_t xbmax, uint32_t ybmax, float x, float y)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    x = __builtin_amdgcn_fmed3f(x, 0.0f, wmax); // clamp to [0, w-1] (interpolation/mod.rs:470-471)
    y = __builtin_amdgcn_fmed3f(y, 0.0f, hmax);
    const uint32_t xb = umin((uint32_t)x, xbmax), yb = umin((uint32_t)y, ybmax); // (uint32_t): truncation = floor, x >= 0
    const float xf = x - (float)xb, yf = y - (float)yb;
    const float nxf = 1.0f - xf, nyf = 1.0f - yf;
    const uint32_t off = __umul24(yb, row_bytes) + xb * 4u;
    const u32x2 r0 = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
    const u32x2 r1 = __builtin_amdgcn_raw_buffer_load_b64(rs, off, row_bytes, 0); // next row: scalar offset, always in range
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float top = lerp_mode<MODE>(ch_f32(r0.x, c), ch_f32(r0.y, c), xf, nxf);
        const float bottom = lerp_mode<MODE>(ch_f32(r1.x, c), ch_f32(r1.y, c), xf, nxf);
        r[c] = floorf(lerp_mode<MODE>(top, bottom, yf, nyf)); // `value as u8`: 0 <= value <= 255 (+ an ulp in FMA mode)
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// One output pixel at (xfl, yfl) with flow f (delta A -> B): A sampled at p - t f, B at p + (1 - t) f, each sample truncated to u8 as
// sample_frame returns it, then the blend of the two truncated samples with the CPU's three roundings in BOTH modes: with integer
// operands and a t like 0.3 a tenth of the exact results are integers themselves (0.7 * 10 + 0.3 * 20 = 13), and there the truncation
// turns any other rounding sequence into a count of difference (measured: 0.25 % of the samples with a fused blend).
// tv / nt / wmax / hmax: per-lane copies of wave-uniform constants (scalar operands halve the VALU issue rate on gfx950).
template <int MODE>
__device__ __forceinline__ uint32_t warp_blend_pixel(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, uint32_t row_bytes, float wmax,
                                                     float hmax, uint32_t xbmax, uint32_t ybmax, float xfl, float yfl, float2 f, float tv,
                                                     float nt)
{
    float ax, ay, bx, by;
    if (MODE == kWarpFma) {
        ax = __builtin_fmaf(-tv, f.x, xfl), ay = __builtin_fmaf(-tv, f.y, yfl);
        bx = __builtin_fmaf(nt, f.x, xfl), by = __builtin_fmaf(nt, f.y, yfl);
    } else {
        ax = xfl - tv * f.x, ay = yfl - tv * f.y;
        bx = xfl + nt * f.x, by = yfl + nt * f.y;
    }
    const float4 sa = sample_corner<MODE>(ra, row_bytes, wmax, hmax, xbmax, ybmax, ax, ay);
    const float4 sb = sample_corner<MODE>(rb, row_bytes, wmax, hmax, xbmax, ybmax, bx, by);
    uint32_t p = 0;
    p = pack_trunc_u8(nt * sa.x + tv * sb.x, 0, p);
    p = pack_trunc_u8(nt * sa.y + tv * sb.y, 1, p);
    p = pack_trunc_u8(nt * sa.z + tv * sb.z, 2, p);
    p = pack_trunc_u8(nt * sa.w + tv * sb.w, 3, p);
    return p;
}

} // namespace

} // namespace nus

// nus_k_interp.hip -- two-frame warp + blend and the BGRA -> RGBA swizzle.
//   warp+blend nu_scaler_core/src/shaders/warp_blend.wgsl:18-47 (geometry),
//              nu_scaler_core/src/interpolation/mod.rs:386-411, :467-510 (rounding)
#include "nus_device.hpp"

#include <hip/hip_fp16.h>

namespace nus {

namespace {

// ---------------------------------------------------------------------------------
// Warp + blend
// ---------------------------------------------------------------------------------

// Zero flow (the live reference behaviour, wgpu_interpolator.rs:275-295): sample
// positions are the pixel centres, so the bilinear samples are the pixels themselves
// and the kernel is a streaming blend, 4 pixels (16 B) per lane.

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

// Same sample through a buffer resource of one frame: 32-bit offsets (one v_mad_u32_u24 per row instead of
// 64-bit address arithmetic per texel) and ONE 8-byte load per row for the horizontal pair -- the pair starts
// at min(x0, w-2), so at the right border (x0 == x1 == w-1) both texels are its second half.
__device__ __forceinline__ float4 sample_trunc_pairs(__amdgpu_buffer_rsrc_t rs, uint32_t w, uint32_t h, float x, float y)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    const float xfl = floorf(x), yfl = floorf(y);
    const uint32_t x0 = (uint32_t)xfl, y0 = (uint32_t)yfl;
    const uint32_t y1 = umin(y0 + 1, h - 1);
    const float xf = x - xfl, yf = y - yfl;
    const float nxf = 1.0f - xf, nyf = 1.0f - yf;
    const uint32_t xb = umin(x0, w - 2); // w >= 2 (checked by the caller)
    const u32x2 r0 = __builtin_amdgcn_raw_buffer_load_b64(rs, (y0 * w + xb) * 4u, 0, 0);
    const u32x2 r1 = __builtin_amdgcn_raw_buffer_load_b64(rs, (y1 * w + xb) * 4u, 0, 0);
    const bool first = x0 == xb; // else x0 == x1 == w-1: both are the pair's second texel
    const uint32_t p00 = first ? r0.x : r0.y, p01 = r0.y, p10 = first ? r1.x : r1.y, p11 = r1.y;
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
// PAIRS: the frames are at least 2 pixels wide and < 4 GiB (host-checked): buffer-resource sampling above.
// HALF: the flow field is 2 x f16 per pixel -- the Rg16Float texture the reference's live path binds
// (wgpu_interpolator.rs:276) -- widened to f32 on load (exact), then the same arithmetic: half the flow bytes.
#ifndef NUS_WARP_ROWS
#define NUS_WARP_ROWS 4 // pixels per thread (rows y, y + 4, y + 8, y + 12 of the block's 16): independent chains of flow load -> gathers -> blend, their loads in flight together
#endif
template <bool PAIRS, bool HALF>
__global__ __launch_bounds__(256) void k_warp_blend_flow(
    const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, const void *__restrict__ flow,
    uint8_t *__restrict__ out, size_t a_stride, size_t b_stride, uint32_t w, uint32_t h, float t, uint32_t sel)
{
    constexpr int R = NUS_WARP_ROWS;
    const uint32_t ybase = __builtin_amdgcn_readfirstlane(blockIdx.y * (4 * R) + threadIdx.y);
    const uint32_t x = blockIdx.x * kWave + threadIdx.x;
    if (ybase >= h || x >= w) return;
    const size_t npx = (size_t)w * h;
    const uint8_t *fa = a + (size_t)blockIdx.z * a_stride, *fb = b + (size_t)blockIdx.z * b_stride;
    float tv = t; // per-lane copy: scalar operands halve the VALU issue rate on gfx950
    asm volatile("" : "+v"(tv));
    const float nt = 1.0f - tv;
    float2 f[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { // the flow vectors of all the thread's pixels first
        const uint32_t y = umin(ybase + 4 * k, h - 1);
        const size_t idx = (size_t)y * w + x;
        if (HALF) {
            const __half2 hf = reinterpret_cast<const __half2 *>(flow)[(size_t)blockIdx.z * npx + idx];
            f[k] = make_float2(__low2float(hf), __high2float(hf));
        } else {
            f[k] = reinterpret_cast<const float2 *>(flow)[(size_t)blockIdx.z * npx + idx];
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t y = ybase + 4 * k;
        if (y >= h) break; // wave-uniform
        const float ax = (float)x - tv * f[k].x, ay = (float)y - tv * f[k].y;
        const float bx = (float)x + nt * f[k].x, by = (float)y + nt * f[k].y;
        float4 sa, sb;
        if (PAIRS) {
            const uint32_t frame_bytes = (uint32_t)(npx * 4);
            sa = sample_trunc_pairs(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fa), 0, frame_bytes, 0x00020000), w, h, ax, ay);
            sb = sample_trunc_pairs(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fb), 0, frame_bytes, 0x00020000), w, h, bx, by);
        } else {
            sa = sample_trunc(reinterpret_cast<const uint32_t *>(fa), w, h, ax, ay);
            sb = sample_trunc(reinterpret_cast<const uint32_t *>(fb), w, h, bx, by);
        }
        uint32_t o = 0;
        o = pack_trunc_u8(nt * sa.x + tv * sb.x, 0, o);
        o = pack_trunc_u8(nt * sa.y + tv * sb.y, 1, o);
        o = pack_trunc_u8(nt * sa.z + tv * sb.z, 2, o);
        o = pack_trunc_u8(nt * sa.w + tv * sb.w, 3, o);
        reinterpret_cast<uint32_t *>(out)[(size_t)blockIdx.z * npx + (size_t)y * w + x] = swz(o, sel);
    }
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

} // namespace

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
            const dim3 block(kWave, 4), grid(cdiv(L.w, 64), cdiv(L.h, 4 * NUS_WARP_ROWS), n);
            const bool pairs = L.w >= 2 && npx * 4 < (1ull << 32);
            const void *fl = reinterpret_cast<const uint8_t *>(L.flow) + (size_t)done * npx * (L.flow_half ? 4 : 8);
#define NUS_WB(P, H) hipLaunchKernelGGL((k_warp_blend_flow<P, H>), grid, block, 0, L.stream, a, b, fl, out, L.a_stride, L.b_stride, L.w, L.h, L.t, L.in_sel)
            if (pairs && L.flow_half) NUS_WB(true, true);
            else if (pairs) NUS_WB(true, false);
            else if (L.flow_half) NUS_WB(false, true);
            else NUS_WB(false, false);
#undef NUS_WB
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += n;
    }
    return hipSuccess;
}

} // namespace nus

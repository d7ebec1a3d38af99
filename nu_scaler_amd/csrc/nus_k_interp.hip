// nus_k_interp.hip -- two-frame warp + blend and the BGRA -> RGBA swizzle.
//   warp+blend nu_scaler_core/src/shaders/warp_blend.wgsl:18-47 (geometry),
//              nu_scaler_core/src/interpolation/mod.rs:386-411, :467-510 (rounding)
#include "nus_device.hpp"
#include "nus_warp_device.hpp"

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
        store_out16<false>(po + i, swz4(make_uint4(blend_px(va.x, vb.x, t, nt), blend_px(va.y, vb.y, t, nt),
                                                             blend_px(va.z, vb.z, t, nt), blend_px(va.w, vb.w, t, nt)), sel));
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

// ---------------------------------------------------------------------------------
// Warp + blend with a dense flow field, rewritten in round 3 around the instruction count (the kernel is bound by SIMD
// issue: ~180 VALU instructions per pixel in the form above, 0.47-0.51 of the HBM roofline).
//
// One bilinear sample, frames of at least 2 x 2 pixels seen through a buffer resource (32-bit offsets):
//  * the texel pair starts at xb = min(x0, w - 2) and the row pair at yb = min(y0, h - 2), so all four texels are always
//    inside the frame -- two 8-byte loads, the second one row_bytes further through the instruction's scalar offset, no
//    border selects -- and the fractions are taken against THAT corner: xf = x - xb, yf = y - yb.  Away from the border
//    these are the reference's fractions (x - floor(x)); at the right / bottom border, where the reference has x0 = x1 =
//    w - 1 and fraction 0 (interpolation/mod.rs:474-483), they are exactly 1 and select the same texel:
//    p(w-2) * (1 - 1) + p(w-1) * 1 = p(w-1), bit for bit in either mode;
//  * MODE_EXACT: every product and sum separately, the CPU's roundings (bit-exact against the oracle);
//    MODE_FMA  : each of the three lerps of a sample as one multiply and one fused multiply-add, fma(b, f, a (1 - f)): 24
//    instead of 36 operations per sample (and the sample positions as one FMA each), inside the +-1 LSB contract of the
//    interpolation path (measured on noise frames with random flows: < 0.1 % of the samples differ, none by more than
//    one count).  The blend of the two samples is exact in both modes (see the kernel).
// The truncation of each sample to u8 (sample_frame returns u8: interpolation/mod.rs:506) stays in both modes.
// Dense flow (2 x f32 or 2 x f16 per pixel, delta A -> B): A sampled at p - t*flow, B at p + (1-t)*flow
// (warp_blend.wgsl:36-37 in texel space).  A thread owns XV consecutive pixels in each of RV rows (rows y, y + 4, .. of the
// block's 4 RV): its flow vectors are loaded first, then the gathers of its pixels -- independent chains -- are in flight
// together.  XV = 2 (frame width even): the two flow vectors of a row are one 16-byte (8-byte) load, the two pixels one
// 8-byte store, and a wave's gathers stay on few cache lines.  blockDim = (64, 4): a block covers 64 XV x 4 RV pixels.
// HALF: the flow field is the Rg16Float texture of the reference's live path (wgpu_interpolator.rs:276), widened exactly.
// Measured on one box, 1080p, us per pair, EXACT / FMA (profiles/r03_warp_kernel_layouts_ab.txt): the round-2 kernel (1 x 4, border
// selects) 10.7 / -; 1 x 4 9.6-10.4 / 8.5-8.7; 4 x 1 11.2-13.1 / 11.1-13.1 (a wave's gathers spread over four times the
// cache lines: the kernel stops being bound by its arithmetic); 2 x 2: 8.7-9.3 / 7.9-8.5.
#ifndef NUS_WARP_XV
#define NUS_WARP_XV 2
#endif
#ifndef NUS_WARP_RV
#define NUS_WARP_RV 2
#endif
template <int MODE, bool HALF, int XV, int RV>
__global__ __launch_bounds__(256) void k_warp_blend_flow(
    const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, const void *__restrict__ flow,
    uint8_t *__restrict__ out, size_t a_stride, size_t b_stride, uint32_t w, uint32_t h, float t, uint32_t sel)
{
    const uint32_t ybase = __builtin_amdgcn_readfirstlane(blockIdx.y * (4 * RV) + threadIdx.y);
    const uint32_t x0 = (blockIdx.x * kWave + threadIdx.x) * XV;
    if (ybase >= h || x0 >= w) return;
    const size_t npx = (size_t)w * h;
    const uint32_t frame_bytes = (uint32_t)(npx * 4), row_bytes = w * 4;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a + (size_t)blockIdx.z * a_stride), 0, frame_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(b + (size_t)blockIdx.z * b_stride), 0, frame_bytes, 0x00020000);
    // per-lane copies of the wave-uniform constants: scalar operands halve the VALU issue rate on gfx950
    float tv = t, wmax = (float)(w - 1), hmax = (float)(h - 1);
    asm volatile("" : "+v"(tv), "+v"(wmax), "+v"(hmax));
    const float nt = 1.0f - tv;
    const size_t frame0 = (size_t)blockIdx.z * npx;
    float2 f[RV][XV];
#pragma unroll
    for (int j = 0; j < RV; ++j) { // the flow vectors of all the thread's pixels first (rows past the frame: the last row's)
        const size_t idx = frame0 + (size_t)umin(ybase + 4 * j, h - 1) * w + x0;
        if (XV == 4) {
            if (HALF) {
                const uint4 raw = *reinterpret_cast<const uint4 *>(reinterpret_cast<const __half2 *>(flow) + idx);
                const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const __half2 hf = *reinterpret_cast<const __half2 *>(&rw[i]);
                    f[j][i] = make_float2(__low2float(hf), __high2float(hf));
                }
            } else {
                const float4 lo = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(flow) + idx)[0];
                const float4 hi = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(flow) + idx)[1];
                f[j][0] = make_float2(lo.x, lo.y);
                f[j][1] = make_float2(lo.z, lo.w);
                f[j][2] = make_float2(hi.x, hi.y);
                f[j][3] = make_float2(hi.z, hi.w);
            }
        } else if (XV == 2) { // (w even, host-checked: the pair is 16-byte / 8-byte aligned)
            if (HALF) {
                const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __half2 *>(flow) + idx);
                const __half2 h0 = *reinterpret_cast<const __half2 *>(&raw.x), h1 = *reinterpret_cast<const __half2 *>(&raw.y);
                f[j][0] = make_float2(__low2float(h0), __high2float(h0));
                f[j][1] = make_float2(__low2float(h1), __high2float(h1));
            } else {
                const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(flow) + idx);
                f[j][0] = make_float2(v.x, v.y);
                f[j][1] = make_float2(v.z, v.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                if (HALF) {
                    const __half2 hf = reinterpret_cast<const __half2 *>(flow)[idx + i];
                    f[j][i] = make_float2(__low2float(hf), __high2float(hf));
                } else {
                    f[j][i] = reinterpret_cast<const float2 *>(flow)[idx + i];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RV; ++j) {
        const uint32_t y = ybase + 4 * j;
        if (y >= h) break; // wave-uniform
        const float yfl = (float)y;
        uint32_t o[XV];
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const uint32_t p = warp_blend_pixel<MODE>(ra, rb, row_bytes, wmax, hmax, w - 2, h - 2, (float)(x0 + i), yfl, f[j][i], tv, nt);
            o[i] = swz(p, sel);
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(out) + frame0 + (size_t)y * w + x0;
        if (XV == 4) {
            store_out16<false>(dst, make_uint4(o[0], o[1], o[2], o[3]));
        } else if (XV == 2) {
            store_out8<false>(dst, make_uint2(o[0], o[1]));
        } else {
#pragma unroll
            for (int i = 0; i < XV; ++i) dst[i] = o[i];
        }
    }
}

// Frames narrower or lower than 2 pixels, or of 4 GiB and more: one pixel per thread, 64-bit addressing (EXACT arithmetic).
__global__ __launch_bounds__(256) void k_warp_blend_flow_tiny(
    const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, const void *__restrict__ flow, bool half,
    uint8_t *__restrict__ out, size_t a_stride, size_t b_stride, uint32_t w, uint32_t h, float t, uint32_t sel)
{
    const uint32_t y = blockIdx.y * 4 + threadIdx.y, x = blockIdx.x * kWave + threadIdx.x;
    if (y >= h || x >= w) return;
    const size_t npx = (size_t)w * h, idx = (size_t)blockIdx.z * npx + (size_t)y * w + x;
    const uint32_t *fa = reinterpret_cast<const uint32_t *>(a + (size_t)blockIdx.z * a_stride);
    const uint32_t *fb = reinterpret_cast<const uint32_t *>(b + (size_t)blockIdx.z * b_stride);
    float2 f;
    if (half) {
        const __half2 hf = reinterpret_cast<const __half2 *>(flow)[idx];
        f = make_float2(__low2float(hf), __high2float(hf));
    } else {
        f = reinterpret_cast<const float2 *>(flow)[idx];
    }
    const float nt = 1.0f - t;
    const float4 sa = sample_trunc(fa, w, h, (float)x - t * f.x, (float)y - t * f.y);
    const float4 sb = sample_trunc(fb, w, h, (float)x + nt * f.x, (float)y + nt * f.y);
    uint32_t o = 0;
    o = pack_trunc_u8(nt * sa.x + t * sb.x, 0, o);
    o = pack_trunc_u8(nt * sa.y + t * sb.y, 1, o);
    o = pack_trunc_u8(nt * sa.z + t * sb.z, 2, o);
    o = pack_trunc_u8(nt * sa.w + t * sb.w, 3, o);
    reinterpret_cast<uint32_t *>(out)[idx] = swz(o, sel);
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
        store_out16<false>(out + i, make_uint4(swap_rb(v.x), swap_rb(v.y), swap_rb(v.z), swap_rb(v.w)));
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
            const void *fl = reinterpret_cast<const uint8_t *>(L.flow) + (size_t)done * npx * (L.flow_half ? 4 : 8);
            const bool corner = L.w >= 2 && L.h >= 2 && npx * 4 < (1ull << 32) && L.w * 4ull < (1u << 24) && L.h < (1u << 24);
            if (!corner) {
                const dim3 block(kWave, 4), grid(cdiv(L.w, 64), cdiv(L.h, 4), n);
                hipLaunchKernelGGL(k_warp_blend_flow_tiny, grid, block, 0, L.stream, a, b, fl, L.flow_half, out, L.a_stride, L.b_stride,
                                   L.w, L.h, L.t, L.in_sel);
            } else {
                constexpr int XV = NUS_WARP_XV, RV = NUS_WARP_RV;
                const uintptr_t align = XV == 4 ? 16 : 8; // of the thread's output pixels (its f16 flow vectors: the same bytes)
                const bool xv_ok = XV == 1 || ((L.w % XV) == 0 && (reinterpret_cast<uintptr_t>(fl) % 16) == 0 &&
                                               (reinterpret_cast<uintptr_t>(out) % align) == 0);
                const dim3 block(kWave, 4), grid(cdiv(L.w, 64 * (xv_ok ? XV : 1)), cdiv(L.h, 4 * RV), n);
#define NUS_WB(M, H, X) hipLaunchKernelGGL((k_warp_blend_flow<M, H, X, RV>), grid, block, 0, L.stream, a, b, fl, out, L.a_stride, L.b_stride, L.w, L.h, L.t, L.in_sel)
#define NUS_WB_X(M, H) do { if (xv_ok) NUS_WB(M, H, XV); else NUS_WB(M, H, 1); } while (0)
                if (L.fma) {
                    if (L.flow_half) NUS_WB_X(kWarpFma, true); else NUS_WB_X(kWarpFma, false);
                } else {
                    if (L.flow_half) NUS_WB_X(kWarpExact, true); else NUS_WB_X(kWarpExact, false);
                }
#undef NUS_WB_X
#undef NUS_WB
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += n;
    }
    return hipSuccess;
}

} // namespace nus

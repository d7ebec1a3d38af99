// nus_k_flow.hip -- optical-flow front end: Gaussian pyramid + Horn-Schunck.
#include "nus_device.hpp"
#include "nus_warp_device.hpp"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace nus {

namespace {

// ---------------------------------------------------------------------------------
// Optical-flow front end: Gaussian pyramid + Horn-Schunck
// ---------------------------------------------------------------------------------
// Images: f32 RGBA, one float4 (16 B) per pixel per lane; flows: float2 per pixel.
// Straight per-pixel kernels with the shaders' exact expression order (no contraction).



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
// the luminance of the blurred level (all Horn-Schunck reads of it, 4 B instead of 16 B per pixel) and, from the same tile, the 2x2-averaged input of the next level.  Each value
// goes through exactly the arithmetic of k_blur<true>, k_blur<false> and k_downsample, so the
// result is bit-identical to the three separate kernels while HBM sees the input once.
constexpr int kPyrTW = 64, kPyrTH = 16;
#ifndef NUS_PYR_THREADS
#define NUS_PYR_THREADS 512
#endif
constexpr int kPyrThreads = NUS_PYR_THREADS; // 3 tiles per CU by LDS: twice the waves hide the LDS round trips of its phases

template <bool U8IN>
__global__ __launch_bounds__(kPyrThreads) void k_pyramid_level(const void *__restrict__ in_all, size_t in_stride,
                                                               float *__restrict__ lum_all, size_t lum_stride,
                                                               float4 *__restrict__ next_all, size_t next_stride, int w, int h)
{
    __shared__ float4 s_a[(kPyrTH + 4) * (kPyrTW + 4)]; // input region; later the V-blurred tile
    __shared__ float4 s_h[(kPyrTH + 4) * kPyrTW];        // H-blurred rows
    constexpr int NW = kPyrThreads / 64;                  // waves per tile: each phase deals its rows to them
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // each XCD takes a contiguous band of tiles: the 2-pixel halos neighbouring tiles share come out of its L2
    const GridPos gp = xcd_contiguous_pos(); // z: frame of the batch; strides in elements (bytes for the RGBA8 input)
    const int bx = (int)gp.x * kPyrTW, by = (int)gp.y * kPyrTH;
    const void *in = U8IN ? static_cast<const void *>(static_cast<const uint8_t *>(in_all) + gp.z * in_stride)
                          : static_cast<const void *>(static_cast<const float4 *>(in_all) + gp.z * in_stride);
    float *level_lum = lum_all + gp.z * lum_stride;
    float4 *next = next_all ? next_all + gp.z * next_stride : nullptr;
    // stage input rows by-2 .. by+17, columns bx-2 .. bx+65, coordinates clamped into the image
    for (int r = ty; r < kPyrTH + 4; r += NW) {
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
    for (int r = ty; r < kPyrTH + 4; r += NW) {
        const float4 *row = s_a + r * (kPyrTW + 4) + tx;
        s_h[r * kPyrTW + tx] = blur5(row[0], row[1], row[2], row[3], row[4]);
    }
    __syncthreads();
    // vertical pass -> blurred level; keep the tile in LDS (s_a is free now) for the downsample
    const int gx = bx + tx;
    for (int r = ty; r < kPyrTH; r += NW) {
        const float4 v = blur5(s_h[r * kPyrTW + tx], s_h[(r + 1) * kPyrTW + tx], s_h[(r + 2) * kPyrTW + tx],
                               s_h[(r + 3) * kPyrTW + tx], s_h[(r + 4) * kPyrTW + tx]);
        s_a[r * kPyrTW + tx] = v;
        const int gy = by + r;
        if (gx < w && gy < h) level_lum[(size_t)gy * w + gx] = lum(v);
    }
    if (next == nullptr) return; // block-uniform
    __syncthreads();
    // 2x2 box average of the blurred tile (tile origin is even, so every 2x2 block is inside it)
    const int ow = (w + 1) / 2, oh = (h + 1) / 2;
    const int dxl = threadIdx.x & 31, dyl = threadIdx.x >> 5; // 32 x 8 outputs per tile
    const int ox = bx / 2 + dxl, oy = by / 2 + dyl;
    if (threadIdx.x < 256 && ox < ow && oy < oh) {
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


// ---- The same pyramid level without LDS ("streamed"), for batches --------------------------------
// A wave owns a strip of 128 columns -- two adjacent columns per lane, the outer lanes are the 2-column
// halo, so 124 columns are written -- and walks down its row block once.  Per input row: both pixels
// converted, the horizontal blur taken from the lane's own two values and the neighbouring lanes' (one
// wave_shr and one wave_shl DPP move per value), the result kept in a five-row register window that the
// vertical blur reads; the 2x2 box average pairs the lane's own two columns of an even and an odd row.
// Rows and columns outside the image are loaded with clamped coordinates, which is exactly what the
// shaders' clamped taps read (gaussian_blur_{h,v}.wgsl:31-40), so no tap is conditional.  Same
// expressions in the same order as k_pyramid_level: identical bits.  HBM traffic as there; no LDS, no barrier.
__device__ __forceinline__ float4 lane_up4(const float4 v)
{
    return make_float4(wave_up(v.x), wave_up(v.y), wave_up(v.z), wave_up(v.w));
}
__device__ __forceinline__ float4 lane_down4(const float4 v)
{
    return make_float4(wave_down(v.x), wave_down(v.y), wave_down(v.z), wave_down(v.w));
}

template <bool U8IN>
__global__ __launch_bounds__(256) void k_pyramid_stream(const void *__restrict__ in_all, size_t in_stride,
                                                        float *__restrict__ lum_all, size_t lum_stride,
                                                        float4 *__restrict__ next_all, size_t next_stride, int w, int h,
                                                        int strips, int row_blocks, int rows_per_block)
{
    constexpr int U = 2 * kWave - 4; // columns a wave writes
    using Raw = typename std::conditional<U8IN, uint32_t, float4>::type;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= strips * row_blocks) return;
    const int rb = g / strips, strip = g - rb * strips;
    const Raw *in = U8IN ? reinterpret_cast<const Raw *>(static_cast<const uint8_t *>(in_all) + blockIdx.y * in_stride)
                         : static_cast<const Raw *>(in_all) + blockIdx.y * in_stride;
    float *level_lum = lum_all + blockIdx.y * lum_stride;
    float4 *next = next_all ? next_all + blockIdx.y * next_stride : nullptr;
    const int xa = strip * U - 2 + 2 * lane, xb = xa + 1; // this lane's columns (xa is even)
    const int ca = clampi(xa, 0, w - 1), cb = clampi(xb, 0, w - 1);
    const bool writer = lane >= 1 && lane <= kWave - 2;
    const int y0 = rb * rows_per_block, y1 = min(y0 + rows_per_block, h); // rows_per_block is even
    const int ow = (w + 1) / 2;

    auto convert = [](Raw p) -> float4 {
        if constexpr (U8IN) return unorm8(p);
        else return p;
    };
    float4 hba[5], hbb[5];   // H-blurred rows r-4 .. r of both columns (slots rotate with the row)
    float4 keep_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), keep_b = keep_a; // the blurred even row of a 2x2 block, until its odd row is there
    Raw na = in[(size_t)clampi(y0 - 2, 0, h - 1) * w + ca], nb = in[(size_t)clampi(y0 - 2, 0, h - 1) * w + cb];

    // input row r arrives in window slot P; from the fifth row on, level row r - 2 leaves
    auto step = [&](int r, auto slot_tag) {
        constexpr int P = decltype(slot_tag)::value;
        const float4 a = convert(na), b = convert(nb);
        {
            const size_t row = (size_t)clampi(r + 1, 0, h - 1) * w; // next row: in flight during this one
            na = in[row + ca], nb = in[row + cb];
        }
        const float4 la = lane_up4(a), lb = lane_up4(b), ra = lane_down4(a), rb4 = lane_down4(b);
        hba[P] = blur5(la, lb, a, b, ra);  // columns xa-2 .. xa+2
        hbb[P] = blur5(lb, a, b, ra, rb4); // columns xb-2 .. xb+2
        const int y = r - 2;
        if (y < y0) return; // the window is still filling (wave-uniform)
        const float4 va = blur5(hba[(P + 1) % 5], hba[(P + 2) % 5], hba[(P + 3) % 5], hba[(P + 4) % 5], hba[P]);
        const float4 vb = blur5(hbb[(P + 1) % 5], hbb[(P + 2) % 5], hbb[(P + 3) % 5], hbb[(P + 4) % 5], hbb[P]);
        if (writer) {
            float *dst = level_lum + (size_t)y * w;
            if (xa < w) dst[xa] = lum(va);
            if (xb < w) dst[xb] = lum(vb);
        }
        if (next == nullptr) return;
        const bool even = (y & 1) == 0;
        if (even) keep_a = va, keep_b = vb; // the upper row of a 2x2 block waits for the lower one
        if (even && y != h - 1) return;     // (odd height: the last row pairs with itself)
        // downsample.wgsl:22-37: (c00 + c10 + c01 + c11) * 0.25, the far column / row clamped into the image
        const bool dup = xb > w - 1;
        const float4 c00 = keep_a, c01 = va;
        const float4 c10 = dup ? keep_a : keep_b, c11 = dup ? va : vb;
        float4 o;
        o.x = (c00.x + c10.x + c01.x + c11.x) * 0.25f;
        o.y = (c00.y + c10.y + c01.y + c11.y) * 0.25f;
        o.z = (c00.z + c10.z + c01.z + c11.z) * 0.25f;
        o.w = (c00.w + c10.w + c01.w + c11.w) * 0.25f;
        if (writer && xa < w) next[(size_t)(y >> 1) * ow + (xa >> 1)] = o;
    };
    const int end = y1 + 1; // last input row taken (clamped into the image by the loads)
    for (int r = y0 - 2; r <= end; r += 5) {
        step(r, std::integral_constant<int, 0>{});
        if (r + 1 > end) break;
        step(r + 1, std::integral_constant<int, 1>{});
        if (r + 2 > end) break;
        step(r + 2, std::integral_constant<int, 2>{});
        if (r + 3 > end) break;
        step(r + 3, std::integral_constant<int, 3>{});
        if (r + 4 > end) break;
        step(r + 4, std::integral_constant<int, 4>{});
    }
}

// ---- The pyramid in FAST arithmetic: luminance only (nus_flow_set_mode(NUS_FLOW_FAST)) -----------------------------------
// Horn-Schunck reads nothing of a level but its luminance (horn_schunck.wgsl:17-20), and blur, box average and luminance are
// all linear: the luminance of the blurred RGBA level is the blurred luminance up to rounding.  This kernel therefore carries
// ONE float per pixel through the pyramid instead of four (level 0: RGBA8 in, luminance plane + the next level's quarter-size
// luminance input out: 18.7 MB per 1080p frame where the exact kernel moves 24.9; a quarter of its arithmetic) with FMAs in
// the 5-tap sums.  Same walk as k_pyramid_stream, FOUR adjacent columns per lane (one 16-byte load and store per row; the
// two halo columns per side are the neighbouring lanes' inner two values, 4 DPP moves per row): a wave writes 248 columns.
// Columns / rows outside the image take the clamped pixel, as the shaders' taps do; waves that touch the left or right image
// border (wave-uniform test) load and store their four columns one by one, the others vector-wide.
__device__ __forceinline__ float blur5_fast(float m2, float m1, float c0, float p1, float p2)
{
    const float W0 = 1.0f / 16.0f, W1 = 4.0f / 16.0f, W2 = 6.0f / 16.0f;
    return __builtin_fmaf(p2, W0, __builtin_fmaf(p1, W1, __builtin_fmaf(c0, W2, __builtin_fmaf(m1, W1, m2 * W0))));
}

#ifndef NUS_PYR_FAST_AHEAD
#define NUS_PYR_FAST_AHEAD 2
#endif
#ifndef NUS_PYR_FAST_NT
#define NUS_PYR_FAST_NT 0 // 1: the luminance plane leaves with non-temporal stores (measured, round 6: see the comment at the store)
#endif
#ifndef NUS_HS_FAST_NT
#define NUS_HS_FAST_NT 0 // bit 0: the flow a launch writes leaves non-temporal; bit 1: the luminance rows and the input flow come in non-temporal
#endif
template <bool U8IN>
__global__ __launch_bounds__(256) void k_pyramid_fast(const void *__restrict__ in_all, size_t in_stride,
                                                      float *__restrict__ lum_all, size_t lum_stride,
                                                      float *__restrict__ next_all, size_t next_stride, int w, int h, int strips,
                                                      int row_blocks, int rows_per_block)
{
#if defined(NUS_ABLATE_PYR_ALIGNED_STRIPS) // timing-only dev build: every lane writes, a wave's row segment is one aligned KiB (the halo
    // columns of lanes 0 and 63 are WRONG: they take their own values) -- what whole-line stores would be worth to this pass
    constexpr int U = 4 * kWave, XOFF = 0;
#else
    constexpr int U = 4 * kWave - 8, XOFF = -4; // columns a wave writes
#endif
    const int lane = threadIdx.x & (kWave - 1);
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= strips * row_blocks) return;
    const int rb = g / strips, strip = g - rb * strips;
    const uint32_t *in8 = U8IN ? reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(in_all) + blockIdx.y * in_stride) : nullptr;
    const float *inf = U8IN ? nullptr : static_cast<const float *>(in_all) + blockIdx.y * in_stride;
    float *level_lum = lum_all + blockIdx.y * lum_stride;
    float *next = next_all ? next_all + blockIdx.y * next_stride : nullptr;
    const int xa = strip * U + XOFF + 4 * lane; // this lane's first column (a multiple of 4)
    const bool writer = XOFF == 0 || (lane >= 1 && lane <= kWave - 2);
    // vector-wide rows need all four columns inside the image (and the wave's other lanes too: wave-uniform choice)
    const bool plain = __builtin_amdgcn_readfirstlane((int)(strip * U + XOFF >= 0 && strip * U + XOFF + 4 * kWave <= w)) != 0;
    int cx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cx[i] = clampi(xa + i, 0, w - 1);
    const int y0 = rb * rows_per_block, y1 = min(y0 + rows_per_block, h); // rows_per_block is even
    const int ow = (w + 1) / 2;
    const float kLum8 = 0.33333f / 255.0f; // (r + g + b) / 255 * 0.33333

    struct Raw {
        uint4 u;
        float4 f;
    };
    // border waves: one vector load at the nearest in-image position (w >= 4), the clamped columns picked out of it afterwards
    const int xv = w >= 4 ? clampi(xa, 0, w - 4) : 0;
    auto fetch = [&](int r) -> Raw {
        Raw v;
        const size_t row = (size_t)clampi(r, 0, h - 1) * w;
        if (plain || w >= 4) {
            const size_t at = row + (plain ? xa : xv);
            // one 16-byte load of a 4-byte-aligned address (a row starts wherever y * w puts it)
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            if constexpr (U8IN) {
                const u32x4_a4 q = *reinterpret_cast<const u32x4_a4 *>(in8 + at);
                v.u = make_uint4(q.x, q.y, q.z, q.w);
            } else {
                const f32x4_a4 q = *reinterpret_cast<const f32x4_a4 *>(inf + at);
                v.f = make_float4(q.x, q.y, q.z, q.w);
            }
        } else {
            if constexpr (U8IN) v.u = make_uint4(in8[row + cx[0]], in8[row + cx[1]], in8[row + cx[2]], in8[row + cx[3]]);
            else v.f = make_float4(inf[row + cx[0]], inf[row + cx[1]], inf[row + cx[2]], inf[row + cx[3]]);
        }
        return v;
    };
    auto pick = [](const float (&l)[4], int i) { return i == 0 ? l[0] : (i == 1 ? l[1] : (i == 2 ? l[2] : l[3])); };
    auto lum8 = [&](uint32_t p) { return ((ch_f32(p, 0) + ch_f32(p, 1)) + ch_f32(p, 2)) * kLum8; };
    auto convert = [&](const Raw &v, float (&l)[4]) {
        if constexpr (U8IN) l[0] = lum8(v.u.x), l[1] = lum8(v.u.y), l[2] = lum8(v.u.z), l[3] = lum8(v.u.w);
        else l[0] = v.f.x, l[1] = v.f.y, l[2] = v.f.z, l[3] = v.f.w;
    };
    float hb[5][4];  // H-blurred rows r-4 .. r of the lane's four columns (slots rotate with the row)
    float keep[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // the blurred even row of a 2x2 block, until its odd row is there
    constexpr int AH = NUS_PYR_FAST_AHEAD; // rows in flight per lane
    Raw nx[AH];
#pragma unroll
    for (int d = 0; d < AH; ++d) nx[d] = fetch(y0 - 2 + d);

    auto step = [&](int r, auto slot_tag) {
        constexpr int P = decltype(slot_tag)::value;
        float l[4];
        convert(nx[0], l);
#pragma unroll
        for (int d = 0; d + 1 < AH; ++d) nx[d] = nx[d + 1];
        nx[AH - 1] = fetch(r + AH); // in flight during this row and the next AH - 1
        if (!plain && w >= 4) { // (wave-uniform) the loaded columns are xv .. xv + 3; this lane's are the clamped cx[]
            const float q[4] = {l[0], l[1], l[2], l[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = pick(q, cx[i] - xv);
        }
        const float m2 = wave_up(l[2]), m1 = wave_up(l[3]), p1 = wave_down(l[0]), p2 = wave_down(l[1]);
        hb[P][0] = blur5_fast(m2, m1, l[0], l[1], l[2]);
        hb[P][1] = blur5_fast(m1, l[0], l[1], l[2], l[3]);
        hb[P][2] = blur5_fast(l[0], l[1], l[2], l[3], p1);
        hb[P][3] = blur5_fast(l[1], l[2], l[3], p1, p2);
        const int y = r - 2;
        if (y < y0) return; // the window is still filling (wave-uniform)
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = blur5_fast(hb[(P + 1) % 5][i], hb[(P + 2) % 5][i], hb[(P + 3) % 5][i], hb[(P + 4) % 5][i], hb[P][i]);
#if defined(NUS_ABLATE_PYR_NO_LUM0_STORE) // timing-only dev build (profiles/r06_flow_level0_luminance_in_jacobi_ablation.txt): the level-0
        // pass without its luminance-plane store -- the most a Jacobi kernel that forms level 0's luminance itself could save here
        if (writer && !U8IN) {
#else
        if (writer) {
#endif
            float *dst = level_lum + (size_t)y * w;
            if (plain) {
#if NUS_PYR_FAST_NT // the plane is read again only after the coarser levels have been solved: nothing to keep in the caches
                typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                f32x4_a4 q;
                q.x = v[0], q.y = v[1], q.z = v[2], q.w = v[3];
                __builtin_nontemporal_store(q, reinterpret_cast<f32x4_a4 *>(dst + xa));
#else
                dst[xa] = v[0], dst[xa + 1] = v[1], dst[xa + 2] = v[2], dst[xa + 3] = v[3]; // (merged into one 16-byte store where aligned)
#endif
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xa + i >= 0 && xa + i < w) dst[xa + i] = v[i];
            }
        }
        if (next == nullptr) return;
        const bool even = (y & 1) == 0;
        if (even) {
#pragma unroll
            for (int i = 0; i < 4; ++i) keep[i] = v[i]; // the upper row of a 2x2 block waits for the lower one
        }
        if (even && y != h - 1) return; // (odd height: the last row pairs with itself)
        // downsample.wgsl:22-37: the 2x2 mean of the BLURRED level; its far column clamps to the level's last column (a lane's
        // value for a column outside the image is the blur at a virtual position, not that pixel)
        const bool d0 = xa + 1 > w - 1, d1 = xa + 3 > w - 1;
        const float o0 = ((keep[0] + (d0 ? keep[0] : keep[1])) + (v[0] + (d0 ? v[0] : v[1]))) * 0.25f;
        const float o1 = ((keep[2] + (d1 ? keep[2] : keep[3])) + (v[2] + (d1 ? v[2] : v[3]))) * 0.25f;
        if (writer && xa >= 0) {
            float *dst = next + (size_t)(y >> 1) * ow + (xa >> 1);
            if (xa < w) dst[0] = o0;
            if (xa + 2 < w) dst[1] = o1;
        }
    };
    const int end = y1 + 1; // last input row taken (clamped into the image by the loads)
    for (int r = y0 - 2; r <= end; r += 5) {
        step(r, std::integral_constant<int, 0>{});
        if (r + 1 > end) break;
        step(r + 1, std::integral_constant<int, 1>{});
        if (r + 2 > end) break;
        step(r + 2, std::integral_constant<int, 2>{});
        if (r + 3 > end) break;
        step(r + 3, std::integral_constant<int, 3>{});
        if (r + 4 > end) break;
        step(r + 4, std::integral_constant<int, 4>{});
    }
}

// Derivatives of one pyramid level, computed once per level instead of once per Jacobi step:
// (ix, iy, it) with exactly the expressions of horn_schunck.wgsl:58-82, 12 bytes per cell.  The
// denominator lambda + ix*ix + iy*iy and its reciprocal are recomputed from them when a tile is
// loaded (same expression, same rounding): cheaper than reading 8 more bytes per cell and halo cell.
// IMG = float4 (RGBA level images) or float (luminance planes written by k_pyramid_level).
__device__ __forceinline__ float lum_of(const float4 *img, size_t i) { return lum(img[i]); }
__device__ __forceinline__ float lum_of(const float *img, size_t i) { return img[i]; }

template <typename IMG>
__device__ __forceinline__ void hs_prepare_cell(const IMG *__restrict__ i1, const IMG *__restrict__ i2, float *__restrict__ coef,
                                                int x, int y, int w, int h)
{
    const int xp = min(x + 1, w - 1), xm = max(x, 1) - 1, yp = min(y + 1, h - 1), ym = max(y, 1) - 1;
    const float ix = (lum_of(i1, (size_t)y * w + xp) - lum_of(i1, (size_t)y * w + xm)) * 0.5f;
    const float iy = (lum_of(i1, (size_t)yp * w + x) - lum_of(i1, (size_t)ym * w + x)) * 0.5f;
    const float it = lum_of(i2, (size_t)y * w + x) - lum_of(i1, (size_t)y * w + x);
    float *c = coef + ((size_t)y * w + x) * 3;
    c[0] = ix;
    c[1] = iy;
    c[2] = it;
}

template <typename IMG>
__global__ __launch_bounds__(256) void k_hs_prepare(const IMG *__restrict__ i1, const IMG *__restrict__ i2, size_t img_stride,
                                                    float *__restrict__ coef, size_t coef_stride, int w, int h)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    hs_prepare_cell(i1 + blockIdx.z * img_stride, i2 + blockIdx.z * img_stride, coef + blockIdx.z * coef_stride, x, y, w, h);
}

// K Jacobi steps per launch on an LDS tile (temporal blocking): a 32x32 output tile is loaded
// with a K-cell halo of the current flow and the per-cell coefficients; step j updates the
// cells whose 3x3 neighbourhood was valid after step j-1 (the region shrinks by one ring per
// step, except at the image border where neighbours clamp inwards), ping-ponging between two
// LDS flow buffers.  Same arithmetic and order as k_horn_schunck, so K launches of that
// kernel and one launch of this one produce identical bits.

struct HsCell {
    float ix, iy, it, den, zinv;
    bool plain_div; // mantissa of den all ones: takes the true division.  Not needed for exactness (tests/test_div_by_recip.py),
                    // but this kernel is 7 % faster with the branch in its step than without (profiles/r02_flow_jacobi_streamed_ab.txt)
};

// T x T output tile, K Jacobi steps per launch (temporal blocking).  A thread owns a vertical run
// of N cells in one column of the (T+2K)^2 tile (R columns x NT/R runs): the 3 x (N+2) flow values
// its cells' neighbourhoods cover are read from LDS once per step and shared between the N cells
// (3(N+2)/N reads per cell instead of 9), lanes of a wave walk consecutive columns (consecutive
// 8-byte LDS words), and nothing in a step is conditional except the final write, so the N sum
// chains interleave.  Each thread owns the same cells in every step: their coefficients live in
// registers and only the two ping-pong flow tiles are in LDS.  Step j may write the cells whose
// distance to the tile edge ("ring") is >= j -- their 3x3 neighbourhood was valid after step j-1;
// at the image border neighbours clamp inwards exactly as in k_horn_schunck.
template <int T, int K, int NT>
__global__ __launch_bounds__(NT) void k_hs_tiled(const float *__restrict__ coef_all, size_t coef_stride, float lambda,
                                                  const float2 *__restrict__ fin_all, size_t fin_stride,
                                                  float2 *__restrict__ fout_all, size_t fout_stride, int w, int h)
{
    constexpr int R = T + 2 * K, RUNS = NT / R, N = (R + RUNS - 1) / RUNS;
    static_assert(RUNS >= 1 && RUNS * N >= R, "tile does not fit the block");
    __shared__ float2 s_flow[2][R * R];
    const int tid = threadIdx.x;
    const int run = tid / R, lx = tid - run * R, ly0 = run * N;
    // each XCD works through a contiguous band of tiles, so the halo cells neighbouring tiles both load come out of its L2
    const GridPos gp = xcd_contiguous_pos(); // z: pair of the batch; strides in elements
    const int tile_x = (int)gp.x, tile_y = (int)gp.y;
    const float *coef = coef_all + gp.z * coef_stride;
    const float2 *fin = fin_all ? fin_all + gp.z * fin_stride : nullptr;
    float2 *fout = fout_all + gp.z * fout_stride;
    const int x0 = tile_x * T - K, y0 = tile_y * T - K; // image coords of LDS cell (0,0)
    // tiles whose loaded region lies strictly inside the image need no clamping at all
    const bool border = x0 < 0 || y0 < 0 || x0 + R > w || y0 + R > h; // block-uniform
    HsCell cell[N];
    int ring[N]; // -1: never written (slot outside the tile or cell outside the image)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int ly = ly0 + i;
        ring[i] = -1;
        if (run < RUNS && ly < R) {
            const int gx = clampi(x0 + lx, 0, w - 1), gy = clampi(y0 + ly, 0, h - 1);
            const size_t g = (size_t)gy * w + gx;
            const float *c = coef + g * 3;
            const float ix = c[0], iy = c[1], den = lambda + ix * ix + iy * iy;
            cell[i].ix = ix;
            cell[i].iy = iy;
            cell[i].it = c[2];
            cell[i].den = den;
            cell[i].zinv = 1.0f / den; // correctly rounded reciprocal, for div_by_recip
            cell[i].plain_div = (__float_as_uint(den) & 0x7fffffu) == 0x7fffffu;
            s_flow[0][ly * R + lx] = fin ? fin[g] : make_float2(0.0f, 0.0f); // null = start from zero flow
            if (gx == x0 + lx && gy == y0 + ly) ring[i] = min(min(lx, ly), min(R - 1 - lx, R - 1 - ly));
        } else {
            cell[i] = HsCell{0.0f, 0.0f, 0.0f, 1.0f, 1.0f, false};
        }
    }
    // LDS columns / rows of the neighbourhood: clamped to the image (border tiles) and to the tile
    // (only cells of ring 0, which are never written, and idle slots are affected by the latter)
    const int gxc = clampi(x0 + lx, 0, w - 1);
    int col[3], row[N + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d) col[d] = clampi((border ? clampi(gxc + d - 1, 0, w - 1) - x0 : lx + d - 1), 0, R - 1);
#pragma unroll
    for (int r = 0; r < N + 2; ++r) {
        const int ly = ly0 + r - 1;
        row[r] = clampi((border ? clampi(y0 + ly, 0, h - 1) - y0 : ly), 0, R - 1) * R;
    }
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int j = 1; j <= K; ++j) {
        const float2 *src = s_flow[cur];
        float2 f[N + 2][3];
#pragma unroll
        for (int r = 0; r < N + 2; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d) f[r][d] = src[row[r] + col[d]];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float su = 0.0f, sv = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    su += f[i + dy][d].x;
                    sv += f[i + dy][d].y;
                }
            // sum / count with count == 9 (horn_schunck.wgsl:38-41)
            const float ua = div_by_recip(su, 9.0f, 1.0f / 9.0f), va = div_by_recip(sv, 9.0f, 1.0f / 9.0f);
            const HsCell &c = cell[i];
            const float num = c.ix * ua + c.iy * va + c.it;
            const float common = c.plain_div ? num / c.den : div_by_recip(num, c.den, c.zinv);
            if (ring[i] >= j) s_flow[cur ^ 1][(ly0 + i) * R + lx] = make_float2(ua - common * c.ix, va - common * c.iy);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (ring[i] < K) continue; // the T x T interior that lies inside the image
        const int ly = ly0 + i;
        fout[(size_t)(y0 + ly) * w + (x0 + lx)] = s_flow[cur][ly * R + lx];
    }
}

// flow_upsample.wgsl:27-36 (linear clamp-to-edge sampler in texel space), vectors * scale
__device__ __forceinline__ float2 flow_upsample_cell(const float2 *__restrict__ src, int sw, int sh, int x, int y, int dw, int dh,
                                                     float scale)
{
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
    return r;
}

// ---- K Jacobi steps per launch, pipelined through registers ("streamed") -------------------------
// A wave owns a strip of 64 columns (lane = column; the outer K on each side are halo, so 64 - 2K
// columns are written) and walks down the rows [lo, hi) of its row block ONCE.  The K steps run as
// K stages of a pipeline one row apart: when row r of level j (the flow after j steps; level 0 is the
// input) arrives, stage j+1 can finish row r-1 of level j+1, which is that level's next arrival, and
// so on down the stages in the same pass of the loop.  Per level a lane keeps, of its own column and
// both neighbouring columns, the row before (r-1) and the running sum of the row before that (r-2):
// exactly the operands the 9-term sum of row r-1 still needs, in the order k_horn_schunck adds them
// (rows top to bottom, columns left to right, starting from 0.0f).  The left / right neighbours come
// from the neighbouring lanes (wave_shr / wave_shl DPP moves) once per row and level, the
// coefficients of a row are loaded once per launch and travel down a K-deep register delay line.
// No LDS, no barriers; HBM traffic is 20 B in + 8 B out per cell per launch (plus the halo columns).
//
// Borders.  Rows: a level's first arrival (row lo) also stands in for row lo-1 and, after row hi-1,
// one more "arrival" repeats that row: at the image border this is the clamp of
// horn_schunck.wgsl:30-36, at a row-block border it makes rows wrong that lie in the K-row halo
// (the wrong region grows by one row per stage and reaches row lo+K-1 / hi-K at most).  Columns:
// a lane whose column is the image's first / last takes itself as left / right neighbour; at a strip
// border the wrong values stay inside the K halo lanes the same way.  Same arithmetic and order as
// k_horn_schunck and k_hs_tiled, so all three produce identical bits.
struct HsRow {
    float ul, uc, ur, vl, vc, vr; // one row of a level: this lane's column and both neighbours
};
struct HsCoef {
    float ix, iy, it, den, zinv;
};

struct HsCoarse { // UPS: the flow a level starts from is the coarser level's, upsampled as it is loaded
    const float2 *flow;
    size_t stride;
    int w, h;
    float scale;
};

template <int K, bool LUM, bool UPS>
__global__ __launch_bounds__(256) void k_hs_stream(const float *__restrict__ coef_all, size_t coef_stride, float lambda,
                                                   const float2 *__restrict__ fin_all, size_t fin_stride,
                                                   float2 *__restrict__ fout_all, size_t fout_stride, int w, int h, int strips,
                                                   int row_blocks, int rows_per_block, HsCoarse coarse)
{
    constexpr int U = kWave - 2 * K; // columns a wave writes
    const int lane = threadIdx.x & (kWave - 1);
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); // wave-uniform, in an SGPR
    if (g >= strips * row_blocks) return;
    const int rb = g / strips, strip = g - rb * strips;
    const float *coef = coef_all + blockIdx.y * coef_stride;
    const float2 *fin = fin_all ? fin_all + blockIdx.y * fin_stride : nullptr; // null = start from zero flow
    float2 *fout = fout_all + blockIdx.y * fout_stride;
    const int x = strip * U - K + lane, xc = clampi(x, 0, w - 1);
    const bool self_l = x <= 0, self_r = x >= w - 1;
    const bool writer = lane >= K && lane < kWave - K && x < w;
    const int y0 = rb * rows_per_block, y1 = min(y0 + rows_per_block, h);
    const int lo = max(y0 - K, 0), hi = min(y1 + K, h);

    HsRow prev[K];      // level j: its newest row so far
    float pu[K], pv[K]; // level j: 0 + l + c + r of the row before that one
    HsCoef cf[K + 1];   // cf[d]: coefficients of row t - d (cf[0] is only the way in)

    auto load_flow = [&](int r) -> float2 {
        if constexpr (UPS) // k_flow_upsample's cell, not written out first (flow_upsample.wgsl)
            return flow_upsample_cell(coarse.flow + blockIdx.y * coarse.stride, coarse.w, coarse.h, xc, min(r, hi - 1), w, h, coarse.scale);
        else
            return fin ? fin[(size_t)min(r, hi - 1) * w + xc] : make_float2(0.0f, 0.0f);
    };
    // Where a row's (ix, iy, it) come from.  !LUM: the coefficient planes k_hs_prepare wrote (coef_all: 3 floats per
    // cell).  LUM: the two luminance planes themselves (coef_all = frame 1's plane, frame 2's one coef_stride
    // floats further on, pairs coef_stride apart as well: consecutive frames of a batch) -- hs_prepare_cell's
    // expressions on a three-row window of frame 1 (rows clamped into the image by the loads, the neighbouring
    // columns from the neighbouring lanes), so the 12 B per cell of coefficients are never written or read.
    auto load_coef = [&](int r, float &a, float &b, float &c) {
        if constexpr (LUM) {
            a = coef[(size_t)clampi(r + 1, 0, h - 1) * w + xc];               // frame 1, row r + 1
            b = coef[coef_stride + (size_t)clampi(r, 0, h - 1) * w + xc];     // frame 2, row r
            c = 0.0f;
        } else {
            const float *p = coef + ((size_t)min(r, hi - 1) * w + xc) * 3;
            a = p[0], b = p[1], c = p[2];
        }
    };
    float2 nf = load_flow(lo);
    float nix, niy, nit; // !LUM: row t's (ix, iy, it).  LUM: nix = frame 1 row t + 1, niy = frame 2 row t
    load_coef(lo, nix, niy, nit);
    float l1_above = 0.0f, l1_row = 0.0f; // LUM: frame 1, rows t - 1 and t
    if constexpr (LUM) {
        l1_above = coef[(size_t)clampi(lo - 1, 0, h - 1) * w + xc];
        l1_row = coef[(size_t)lo * w + xc];
    }

    // One pass of the pipeline: row t of the input arrives, every level that has a row to take takes it.
    // STEADY (lo + K <= t < hi): every level has a real arrival and an earlier row -- no conditions.
    auto pass = [&](int t, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const float2 pf = load_flow(t + 1); // next pass's row: in flight during this pass
        float pix, piy, pit;
        load_coef(t + 1, pix, piy, pit);
        {
            float ix = nix, iy = niy, it = nit;
            if constexpr (LUM) { // horn_schunck.wgsl:58-82, as hs_prepare_cell
                const float left = wave_up(l1_row), right = wave_down(l1_row);
                ix = ((self_r ? l1_row : right) - (self_l ? l1_row : left)) * 0.5f;
                iy = (nix - l1_above) * 0.5f;
                it = niy - l1_row;
                l1_above = l1_row, l1_row = nix;
            }
            const float den = lambda + ix * ix + iy * iy;
            cf[0] = HsCoef{ix, iy, it, den, 1.0f / den}; // correctly rounded reciprocal, for div_by_recip
        }
        float2 arr = nf; // level 0's arrival: row t of the input flow
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int r = t - j; // the row arriving at level j in this pass
            if (!STEADY) {
                if (r < lo) break;    // the pipeline is still filling (wave-uniform)
                if (r > hi) continue; // this level is done, deeper ones are draining
            }
            HsRow n;
            if (!STEADY && r == hi) {
                n = prev[j]; // past the last row: it repeats
            } else {
                n.uc = arr.x, n.vc = arr.y;
                const float ul = wave_up(arr.x), ur = wave_down(arr.x), vl = wave_up(arr.y), vr = wave_down(arr.y);
                n.ul = self_l ? arr.x : ul, n.ur = self_r ? arr.x : ur;
                n.vl = self_l ? arr.y : vl, n.vr = self_r ? arr.y : vr;
            }
            if (!STEADY && r == lo) { // first row: also the row above it
                prev[j] = n;
                float su = 0.0f, sv = 0.0f;
                su += n.ul, sv += n.vl, su += n.uc, sv += n.vc, su += n.ur, sv += n.vr;
                pu[j] = su, pv[j] = sv;
                break; // deeper levels have nothing yet
            }
            // row r-1 of level j+1 (horn_schunck.wgsl:26-50): the sum continues over rows r-1 and r
            const HsRow b = prev[j];
            float su = pu[j], sv = pv[j];
            su += b.ul, sv += b.vl, su += b.uc, sv += b.vc, su += b.ur, sv += b.vr;
            su += n.ul, sv += n.vl, su += n.uc, sv += n.vc, su += n.ur, sv += n.vr;
            const float ua = div_by_recip(su, 9.0f, 1.0f / 9.0f), va = div_by_recip(sv, 9.0f, 1.0f / 9.0f);
            const HsCoef c = cf[j + 1]; // row r-1 entered j+1 passes ago
            const float num = c.ix * ua + c.iy * va + c.it;
            // == num / den for every den, the all-ones mantissa included (tests/test_div_by_recip.py): no branch in the pass
            const float common = div_by_recip(num, c.den, c.zinv);
            arr = make_float2(ua - common * c.ix, va - common * c.iy);
            // the row that was newest becomes the row above
            float qu = 0.0f, qv = 0.0f;
            qu += b.ul, qv += b.vl, qu += b.uc, qv += b.vc, qu += b.ur, qv += b.vr;
            pu[j] = qu, pv[j] = qv;
            prev[j] = n;
            if (j == K - 1) {
                const int y = r - 1;
                if (y >= y0 && y < y1 && writer) fout[(size_t)y * w + x] = arr;
            }
        }
#pragma unroll
        for (int d = K; d >= 1; --d) cf[d] = cf[d - 1];
        nf = pf, nix = pix, niy = piy, nit = pit;
    };
    int t = lo;
    for (; t < min(lo + K, hi + K); ++t) pass(t, std::false_type{}); // fill
    for (; t < hi; ++t) pass(t, std::true_type{});
    for (; t < hi + K; ++t) pass(t, std::false_type{}); // drain
}

// ---- The same pipeline in FAST arithmetic (nus_flow_set_mode(NUS_FLOW_FAST)) ------------------------------------------
// k_hs_stream reproduces, operation for operation, arithmetic the reference never executes (compute_coarse_flow is not wired
// into interpolate_py, wgpu_interpolator.rs:1156-1203; no fixture exists): 9-term sums in the shader's order, true divisions
// (as exact reciprocal sequences).  This kernel computes the same Jacobi step in the cheapest f32 form:
//   * the 3x3 sum separably -- the vertical sum of a column's rows r-2, r-1, r first (own column only: a level keeps two rows of
//     (u, v) instead of three columns of them and two running sums), then left + centre + right of that sum across lanes
//     (4 DPP moves per step as before, 8 adds instead of 18); the clamp of horn_schunck.wgsl:30-36 is separable, so the border
//     cells take exactly the cells the shader takes;
//   * the mean as a multiply by 1/9, the update as  u' = ua - num * gx,  gx = ix / (lambda + ix^2 + iy^2)  precomputed per
//     cell when its row enters the pipeline (one v_rcp_f32 per cell and launch), num = fma(ix, ua, fma(iy, va, it)).
// ~22 f32 operations + 9 register moves per cell and step where the exact kernel has ~60.  Contract (tests/test_flow.py): the
// flow of the full estimator within 1e-3 px of orc_flow_estimate's at 1080p, the frame interpolated with it within 1 LSB and
// on < 0.1 % of the samples different.  Always takes the derivatives from the luminance planes (LUM of k_hs_stream).
#ifndef NUS_HS_FAST_AHEAD
#define NUS_HS_FAST_AHEAD 2 // passes between a row's request and its use in k_hs_stream_fast
#endif
typedef float f32x2_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store_f2(float2 v, float2 *p)
{
    f32x2_nt q;
    q.x = v.x, q.y = v.y;
    __builtin_nontemporal_store(q, reinterpret_cast<f32x2_nt *>(p));
}
__device__ __forceinline__ float2 nt_load_f2(const float2 *p)
{
    const f32x2_nt q = __builtin_nontemporal_load(reinterpret_cast<const f32x2_nt *>(p));
    return make_float2(q.x, q.y);
}
struct HsFastCoef {
    float ix, iy, it, gx, gy;
};

// RING (round 5): the same passes with every queue of the pipeline as a RING whose slot indices are compile-time constants -- the
// coefficient delay line (K + 1 rows x 5 floats), each level's two previous rows, the rows in flight from memory.  The shifting
// form above moves each of them down by one register per pass: 5 K + 4 K + 8 v_mov of the ~170 instructions of a 5-step pass, in a
// kernel that is bound by instruction issue.  A pass's slot indices depend on (t - lo) mod R only, R = 6 rows for K <= 5 (a multiple
// of 3 for the three-row rings, >= K + 1 for the delay line, >= 5 for the luminance rows t-1 .. t+3) and 9 rows for K = 6 .. 8, so the
// pass is instantiated R times (x 2: steady / fill-and-drain) as one straight-line turn of the rings: no value ever changes
// register.  Same operations on the same operands in the same order: bit-identical flows (tests/test_flow.py runs both forms).
// K > 8 keeps the shifting form (12 copies of a ten-step pass would not fit the instruction cache).
#ifndef NUS_HS_FAST_RING
#define NUS_HS_FAST_RING 1
#endif
#ifndef NUS_HS_FAST_INNER_STRIPS
#define NUS_HS_FAST_INNER_STRIPS 0 // dev macro: 1 = strips that touch no image border run a copy of the pass without the clamp's selects
                                   // (22 of ~100 VALU instructions of a 5-step pass).  Measured, round 5: identical flows, 110 / 126 / 140 VGPRs
                                   // instead of 95 / 108 / 122 (one wave per SIMD fewer), 42.1 - 42.8 us per 1080p pair against 41.7 - 42.3: not kept
#endif
#ifndef NUS_HS_FAST_RING_MAXK
#define NUS_HS_FAST_RING_MAXK 8 // launches of up to this many steps take the ring form
#endif
// WARP (round 5, ring form only): the launch that finishes the finest level's flow also warps + blends the pair's two frames with it
// (warp_blend_pixel<kWarpFma>, the very code of k_warp_blend_flow) and stores the in-between pixel; the flow itself is stored only
// if the caller wants it (fout_all != nullptr).  Saves the flow's way through HBM to a separate warp launch (33 MB per 1080p pair).
template <int K, bool UPS, bool RING = false, bool WARP = false>
__global__ __launch_bounds__(256) void k_hs_stream_fast(const float *__restrict__ lum_all, size_t lum_stride, float lambda,
                                                        const float2 *__restrict__ fin_all, size_t fin_stride,
                                                        float2 *__restrict__ fout_all, size_t fout_stride, int w, int h, int strips,
                                                        int row_blocks, int rows_per_block, HsCoarse coarse, HsWarp warp)
{
    static_assert(!WARP || (RING && !UPS), "WARP: the ring form of a launch that continues a full-resolution flow");
    constexpr int U = kWave - 2 * K; // columns a wave writes
    const int lane = threadIdx.x & (kWave - 1);
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); // wave-uniform, in an SGPR
    if (g >= strips * row_blocks) return;
    const int rb = g / strips, strip = g - rb * strips;
    const float *lum1 = lum_all + blockIdx.y * lum_stride; // frame 1's plane; frame 2's follows lum_stride floats further on
    const float2 *fin = fin_all ? fin_all + blockIdx.y * fin_stride : nullptr; // null = start from zero flow
    float2 *fout = fout_all + blockIdx.y * fout_stride;
    // WARP: the pair's frames through buffer resources (32-bit offsets: frames < 4 GiB, host-checked), per-lane copies of the constants
    const uint32_t wframe_bytes = (uint32_t)w * (uint32_t)h * 4u;
    const uint8_t *wfa = WARP ? warp.frames + (size_t)blockIdx.y * warp.frame_stride : nullptr;
    const __amdgpu_buffer_rsrc_t wra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(wfa), 0, WARP ? wframe_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(WARP ? wfa + warp.frame_stride : nullptr), 0, WARP ? wframe_bytes : 0u, 0x00020000);
    uint32_t *wmid = WARP ? reinterpret_cast<uint32_t *>(warp.mid) + (size_t)blockIdx.y * (size_t)w * h : nullptr;
    float wtv = warp.t, wwmax = (float)(w - 1), whmax = (float)(h - 1);
    if constexpr (WARP) asm volatile("" : "+v"(wtv), "+v"(wwmax), "+v"(whmax));
    const float wnt = 1.0f - wtv;
    const int x = strip * U - K + lane, xc = clampi(x, 0, w - 1);
    const bool self_l = x <= 0, self_r = x >= w - 1;
    const bool writer = lane >= K && lane < kWave - K && x < w;
    const int y0 = rb * rows_per_block, y1 = min(y0 + rows_per_block, h);
    const int lo = max(y0 - K, 0), hi = min(y1 + K, h);

    static_assert(!RING || K <= 8, "the ring form is instantiated for K <= 8 (R = 6 or 9)");
    constexpr int R = K <= 5 ? 6 : 9; // RING: passes per turn of every ring
    float2 a1[RING ? 1 : K], a2[RING ? 1 : K]; // level j: its rows r-1 and r-2 (this lane's column)
    HsFastCoef cf[RING ? 1 : K + 1];           // cf[d]: coefficients of row t - d (cf[0] is only the way in)
    float2 ar[RING ? K : 1][3];                // RING: level j's rows r, r-1, r-2 in slots (row - lo) mod 3
    HsFastCoef cfr[RING ? R : 1];              // RING: coefficients of row t in slot (t - lo) mod R
    float l1r[RING ? R : 1];                   // RING: frame 1's rows t-1 .. t+3 in slots (row - lo) mod R
    float q2r[3];                              // RING: frame 2's rows t .. t+2
    float2 qfr[3];                             // RING: the input flow's rows t .. t+2 (not UPS)
#if defined(NUS_ABLATE_HS_LUM0_COST)
    float abl_ring[2][R];
#pragma unroll
    for (int i = 0; i < R; ++i) abl_ring[0][i] = abl_ring[1][i] = 0.0f;
#endif

    // UPS: the level starts from the coarser level's flow, sampled as flow_upsample.wgsl:27-36 samples it (linear, clamp to edge,
    // texel space) while it is loaded.  The column part of the sample position is this lane's for the whole launch (the shader's own
    // expressions, once), and so is the horizontal interpolation of a COARSE row: a wave walks down the fine rows, the pair of
    // coarse rows a fine row lies between advances by one every other row -- so a coarse row is loaded (2 texels per lane) and
    // interpolated in x once, kept as `hb` (the lower of the pair) / `ha` (the upper), and the next one is requested when the pair
    // advances, two fine rows before it is needed.  A fine row then costs its row position (an FMA with the ratio of the heights
    // instead of a division) and one interpolation in y.  Loading 4 texels per fine row instead made the launch that opens a
    // level 25 % SLOWER than the ones that read a full-resolution flow, although it moves 25 % fewer bytes.
    const float2 *csrc = UPS ? coarse.flow + blockIdx.y * coarse.stride : nullptr;
    int cx0 = 0, cx1 = 0, cy = 0; // cy (wave-uniform): the coarse row in hb
    float cfx = 0.0f, cry = 0.0f;
    float2 ha = make_float2(0.0f, 0.0f), hb = ha, ra = ha, rb2 = ha; // ra / rb2: texels x0 / x1 of coarse row cy + 1, in flight
    auto coarse_row_pos = [&](int r, int &y0c, int &y1c) -> float { // fine row r -> its coarse rows and the fraction between them
        const float sy = __builtin_fmaf((float)min(r, hi - 1) + 0.5f, cry, -0.5f), fy0 = floorf(sy);
        y0c = __builtin_amdgcn_readfirstlane(clampi((int)fy0, 0, coarse.h - 1));
        y1c = __builtin_amdgcn_readfirstlane(clampi((int)fy0 + 1, 0, coarse.h - 1));
        return sy - fy0;
    };
    auto lerp_x = [&](const float2 a, const float2 b) {
        return make_float2(__builtin_fmaf(cfx, b.x - a.x, a.x), __builtin_fmaf(cfx, b.y - a.y, a.y));
    };
    auto request_coarse = [&](int yc) {
        const float2 *row = csrc + (size_t)min(yc, coarse.h - 1) * coarse.w;
        ra = row[cx0], rb2 = row[cx1];
    };
    if constexpr (UPS) {
        const float u = ((float)xc + 0.5f) / (float)w, sx = u * (float)coarse.w - 0.5f, fx0 = floorf(sx);
        cfx = sx - fx0;
        cx0 = clampi((int)fx0, 0, coarse.w - 1), cx1 = clampi((int)fx0 + 1, 0, coarse.w - 1);
        cry = (float)coarse.h / (float)h;
        int y0c, y1c;
        (void)coarse_row_pos(lo, y0c, y1c);
        cy = y1c;
        request_coarse(y0c);
        ha = lerp_x(ra, rb2);
        request_coarse(y1c);
        hb = lerp_x(ra, rb2);
        request_coarse(cy + 1);
    }
    // the upsampled flow of fine row r (rows are asked for in increasing order)
    auto upsampled_row = [&](int r) -> float2 {
        int y0c, y1c;
        const float fy = coarse_row_pos(r, y0c, y1c);
        if (y1c > cy) { // (wave-uniform) the pair advances: the requested row comes in, the one after it is requested
            ha = hb;
            hb = lerp_x(ra, rb2);
            cy = cy + 1;
            request_coarse(cy + 1);
        }
        const float2 top = y0c == cy ? hb : ha; // (the clamped first / last pair: both rows the same)
        return make_float2(__builtin_fmaf(fy, hb.x - top.x, top.x) * coarse.scale, __builtin_fmaf(fy, hb.y - top.y, top.y) * coarse.scale);
    };
    auto load_flow = [&](int r) -> float2 {
        if constexpr (UPS) {
#if defined(NUS_HS_FAST_UPS_EXACT) // dev macro: the shader's own expressions (bisecting)
            return flow_upsample_cell(csrc, coarse.w, coarse.h, xc, min(r, hi - 1), w, h, coarse.scale);
#else
            return upsampled_row(r);
#endif
        } else {
            if (warp.in_half) { // (wave-uniform) the previous launch of this level stored halves
                const __half2 hv = reinterpret_cast<const __half2 *>(fin_all)[blockIdx.y * fin_stride + (size_t)min(r, hi - 1) * w + xc];
                return __half22float2(hv);
            }
            if (NUS_HS_FAST_NT & 2) return fin ? nt_load_f2(&fin[(size_t)min(r, hi - 1) * w + xc]) : make_float2(0.0f, 0.0f);
            return fin ? fin[(size_t)min(r, hi - 1) * w + xc] : make_float2(0.0f, 0.0f);
        }
    };
    // rows are requested NUS_HS_FAST_AHEAD passes before they are used: a FAST pass is short (~120 instructions), and with one row
    // in flight per wave the launch was bound by memory latency, not by bandwidth or instruction issue
    constexpr int AH = NUS_HS_FAST_AHEAD;
    float2 qf[AH]; // (UPS: unused -- the upsampled row comes out of registers when it is due, see upsampled_row)
    float q1[AH], q2[AH]; // q*[d]: row t + d's flow, frame 1's row t + d + 1, frame 2's row t + d (d = 0: this pass's)
#pragma unroll
    for (int d = 0; d < AH; ++d) {
        if constexpr (!UPS) qf[d] = load_flow(lo + d);
        q1[d] = lum1[(size_t)clampi(lo + d + 1, 0, h - 1) * w + xc];
        q2[d] = lum1[lum_stride + (size_t)clampi(lo + d, 0, h - 1) * w + xc];
    }
    float l1_above = lum1[(size_t)clampi(lo - 1, 0, h - 1) * w + xc], l1_row = lum1[(size_t)lo * w + xc]; // frame 1, rows t - 1, t
    const float ninth = 1.0f / 9.0f;
    if constexpr (RING) {
        static_assert(!RING || NUS_HS_FAST_AHEAD == 2, "the ring form keeps rows t .. t+2 in three-slot rings");
        l1r[R - 1] = l1_above, l1r[0] = l1_row, l1r[1] = q1[0], l1r[2] = q1[1];
        q2r[0] = q2[0], q2r[1] = q2[1];
        if constexpr (!UPS) qfr[0] = qf[0], qfr[1] = qf[1];
    }
    // One pass in ring form, phase P = (t - lo) mod R (compile time)
    auto ring_pass = [&](int t, auto steady_tag, auto phase_tag, auto border_tag) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        // BORDER = false: a strip all of whose 64 columns lie strictly inside the image (every strip but the first and the last) --
        // no lane's neighbour is the lane itself, the selects that implement the shader's clamp drop out of the pass
        constexpr bool BORDER = decltype(border_tag)::value;
        const bool sl = BORDER && self_l, sr = BORDER && self_r;
        constexpr int P = decltype(phase_tag)::value;
        float2 nf;
        if constexpr (UPS) nf = load_flow(t);
        else {
            nf = qfr[P % 3];
            qfr[(P + 2) % 3] = load_flow(t + 2); // in flight during this pass and the next
        }
#if defined(NUS_ABLATE_HS_LUM0_COST) // timing-only dev build (same file): per pass and frame, the instructions an in-kernel level-0
        // luminance costs at least -- RGBA8 -> luminance (3 converts, 2 adds, 1 multiply), the two halo columns per side from the
        // neighbouring lanes (4 DPP moves), the 5-tap row blur, a ring of blurred rows and the 5-tap column blur -- run on values
        // that have already arrived and folded into them as + 0 * result (IEEE: not removable; the flows stay the same).  Not
        // counted: the 4 extra halo lanes per wave such a kernel needs (U = 50 instead of 54 columns at K = 5).
        float row = l1r[P % R];
        const float above = l1r[(P + R - 1) % R], n1 = l1r[(P + 1) % R];
        float n2 = q2r[P % 3];
        if (w > 1000) { // (wave-uniform: level 0 of a 1080p stream only)
            auto cost = [&](float v, float (&ring)[R]) {
                const uint32_t p = __float_as_uint(v);
                const float l = ((ch_f32(p, 0) + ch_f32(p, 1)) + ch_f32(p, 2)) * (0.33333f / 255.0f);
                const float m1 = wave_up(l), p1 = wave_down(l), m2 = wave_up(m1), p2 = wave_down(p1);
                ring[P % R] = blur5_fast(m2, m1, l, p1, p2);
                return blur5_fast(ring[(P + 1) % R], ring[(P + 2) % R], ring[(P + 3) % R], ring[(P + 4) % R], ring[P % R]);
            };
            row = __builtin_fmaf(0.0f, cost(row, abl_ring[0]), row);
            n2 = __builtin_fmaf(0.0f, cost(n2, abl_ring[1]), n2);
        }
#else
        const float row = l1r[P % R], above = l1r[(P + R - 1) % R], n1 = l1r[(P + 1) % R], n2 = q2r[P % 3];
#endif
        if (NUS_HS_FAST_NT & 2) {
            l1r[(P + 3) % R] = __builtin_nontemporal_load(&lum1[(size_t)clampi(t + 3, 0, h - 1) * w + xc]);
            q2r[(P + 2) % 3] = __builtin_nontemporal_load(&lum1[lum_stride + (size_t)clampi(t + 2, 0, h - 1) * w + xc]);
        } else {
            l1r[(P + 3) % R] = lum1[(size_t)clampi(t + 3, 0, h - 1) * w + xc];
            q2r[(P + 2) % 3] = lum1[lum_stride + (size_t)clampi(t + 2, 0, h - 1) * w + xc];
        }
        {
            const float left = wave_up(row), right = wave_down(row);
            const float ix = ((sr ? row : right) - (sl ? row : left)) * 0.5f;
            const float iy = (n1 - above) * 0.5f;
            const float it = n2 - row;
            const float rinv = __builtin_amdgcn_rcpf(__builtin_fmaf(iy, iy, __builtin_fmaf(ix, ix, lambda)));
            cfr[P] = HsFastCoef{ix, iy, it, ix * rinv, iy * rinv};
        }
        float2 arr = nf; // level 0's arrival: row t of the input flow
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int r = t - j; // the row arriving at level j in this pass
            constexpr int kBig = 3 * R;
            const int s0 = (P - j + kBig) % 3, s1 = (P - j - 1 + kBig) % 3, s2 = (P - j - 2 + kBig) % 3; // rows r, r-1, r-2 (constants once unrolled)
            if (!STEADY) {
                if (r < lo) break;    // the pipeline is still filling (wave-uniform)
                if (r > hi) continue; // this level is done, deeper ones are draining
                if (r == hi) arr = ar[j][s1]; // past the last row: it repeats
                if (r == lo) { // first row: also the row above it
                    ar[j][s0] = arr, ar[j][s1] = arr;
                    break; // deeper levels have nothing yet
                }
            }
            const float su = (ar[j][s2].x + ar[j][s1].x) + arr.x, sv = (ar[j][s2].y + ar[j][s1].y) + arr.y;
            const float lu = wave_up(su), ru = wave_down(su), lv = wave_up(sv), rv = wave_down(sv);
            const float ua = (((sl ? su : lu) + su) + (sr ? su : ru)) * ninth;
            const float va = (((sl ? sv : lv) + sv) + (sr ? sv : rv)) * ninth;
            const HsFastCoef c = cfr[(P - j - 1 + kBig) % R]; // row r-1 entered j+1 passes ago
            const float num = __builtin_fmaf(c.ix, ua, __builtin_fmaf(c.iy, va, c.it));
            ar[j][s0] = arr; // the arriving row takes the slot of row r-3, which nothing reads any more
            arr = make_float2(__builtin_fmaf(-num, c.gx, ua), __builtin_fmaf(-num, c.gy, va));
            if (j == K - 1) {
                const int y = r - 1;
                if (y >= y0 && y < y1 && writer) {
                    if (!WARP || fout_all != nullptr) {
                        if (warp.out_half) // (wave-uniform) the level's final flow as Rg16Float
                            reinterpret_cast<__half2 *>(fout_all)[blockIdx.y * fout_stride + (size_t)y * w + x] = __floats2half2_rn(arr.x, arr.y);
                        else if (NUS_HS_FAST_NT & 1)
                            nt_store_f2(arr, &fout[(size_t)y * w + x]);
                        else
                            fout[(size_t)y * w + x] = arr;
                    }
                    if constexpr (WARP) {
                        const uint32_t p = warp_blend_pixel<kWarpFma>(wra, wrb, (uint32_t)w * 4u, wwmax, whmax, (uint32_t)w - 2u, (uint32_t)h - 2u,
                                                                      (float)x, (float)y, arr, wtv, wnt);
                        wmid[(size_t)y * w + x] = swz(p, warp.sel);
                    }
                }
            }
        }
    };
    // A whole turn of the rings (R passes) as straight-line code: every value keeps its register across the loop's back edge.
    auto ring_turn = [&](int t, auto steady_tag, auto border_tag) __attribute__((always_inline)) {
        ring_pass(t, steady_tag, std::integral_constant<int, 0>{}, border_tag);
        ring_pass(t + 1, steady_tag, std::integral_constant<int, 1>{}, border_tag);
        ring_pass(t + 2, steady_tag, std::integral_constant<int, 2>{}, border_tag);
        ring_pass(t + 3, steady_tag, std::integral_constant<int, 3>{}, border_tag);
        ring_pass(t + 4, steady_tag, std::integral_constant<int, 4>{}, border_tag);
        ring_pass(t + 5, steady_tag, std::integral_constant<int, 5>{}, border_tag);
        if constexpr (R == 9) {
            ring_pass(t + 6, steady_tag, std::integral_constant<int, 6>{}, border_tag);
            ring_pass(t + 7, steady_tag, std::integral_constant<int, 7>{}, border_tag);
            ring_pass(t + 8, steady_tag, std::integral_constant<int, 8>{}, border_tag);
        }
    };
    if constexpr (RING) {
        // the first turn fills the pipeline (K <= R - 1 passes of it) and the turns from the first one that reaches row hi on drain it:
        // the general form of the pass, whose levels look at their row number (passes past hi + K find nothing to do); all turns in
        // between are steady
        auto walk = [&](auto border_tag) __attribute__((always_inline)) {
            int t = lo;
            ring_turn(t, std::false_type{}, border_tag);
            for (t += R; t + R <= hi; t += R) ring_turn(t, std::true_type{}, border_tag);
            for (; t < hi + K; t += R) ring_turn(t, std::false_type{}, border_tag);
        };
#if NUS_HS_FAST_INNER_STRIPS
        const bool inner = strip * U - K > 0 && strip * U - K + (kWave - 1) < w - 1; // wave-uniform
        if (inner) walk(std::false_type{});
        else
#endif
            walk(std::true_type{});
        return;
    }

    auto pass = [&](int t, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        float2 nf, pf = make_float2(0.0f, 0.0f);
        if constexpr (UPS) nf = load_flow(t);
        else nf = qf[0], pf = load_flow(t + AH); // in flight during this pass and the next AH - 1
        const float n1 = q1[0], n2 = q2[0];
        const float p1 = lum1[(size_t)clampi(t + AH + 1, 0, h - 1) * w + xc];
        const float p2 = lum1[lum_stride + (size_t)clampi(t + AH, 0, h - 1) * w + xc];
        {
            const float left = wave_up(l1_row), right = wave_down(l1_row);
            const float ix = ((self_r ? l1_row : right) - (self_l ? l1_row : left)) * 0.5f;
            const float iy = (n1 - l1_above) * 0.5f;
            const float it = n2 - l1_row;
            l1_above = l1_row, l1_row = n1;
            const float rinv = __builtin_amdgcn_rcpf(__builtin_fmaf(iy, iy, __builtin_fmaf(ix, ix, lambda)));
            cf[0] = HsFastCoef{ix, iy, it, ix * rinv, iy * rinv};
        }
        float2 arr = nf; // level 0's arrival: row t of the input flow
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int r = t - j; // the row arriving at level j in this pass
            if (!STEADY) {
                if (r < lo) break;    // the pipeline is still filling (wave-uniform)
                if (r > hi) continue; // this level is done, deeper ones are draining
                if (r == hi) arr = a1[j]; // past the last row: it repeats
                if (r == lo) { // first row: also the row above it
                    a1[j] = arr, a2[j] = arr;
                    break; // deeper levels have nothing yet
                }
            }
            // row r-1 of level j+1: vertical sum of rows r-2, r-1, r of this column, then left + centre + right of it
            const float su = (a2[j].x + a1[j].x) + arr.x, sv = (a2[j].y + a1[j].y) + arr.y;
            const float lu = wave_up(su), ru = wave_down(su), lv = wave_up(sv), rv = wave_down(sv);
            const float ua = (((self_l ? su : lu) + su) + (self_r ? su : ru)) * ninth;
            const float va = (((self_l ? sv : lv) + sv) + (self_r ? sv : rv)) * ninth;
            const HsFastCoef c = cf[j + 1]; // row r-1 entered j+1 passes ago
            const float num = __builtin_fmaf(c.ix, ua, __builtin_fmaf(c.iy, va, c.it));
            a2[j] = a1[j], a1[j] = arr;
            arr = make_float2(__builtin_fmaf(-num, c.gx, ua), __builtin_fmaf(-num, c.gy, va));
            if (j == K - 1) {
                const int y = r - 1;
                if (y >= y0 && y < y1 && writer) {
                    if (warp.out_half)
                        reinterpret_cast<__half2 *>(fout_all)[blockIdx.y * fout_stride + (size_t)y * w + x] = __floats2half2_rn(arr.x, arr.y);
                    else
                        fout[(size_t)y * w + x] = arr;
                }
            }
        }
#pragma unroll
        for (int d = K; d >= 1; --d) cf[d] = cf[d - 1];
#pragma unroll
        for (int d = 0; d + 1 < AH; ++d) {
            if constexpr (!UPS) qf[d] = qf[d + 1];
            q1[d] = q1[d + 1], q2[d] = q2[d + 1];
        }
        if constexpr (!UPS) qf[AH - 1] = pf;
        q1[AH - 1] = p1, q2[AH - 1] = p2;
    };
    int t = lo;
    for (; t < min(lo + K, hi + K); ++t) pass(t, std::false_type{}); // fill
    for (; t < hi; ++t) pass(t, std::true_type{});
    for (; t < hi + K; ++t) pass(t, std::false_type{}); // drain
}

__global__ __launch_bounds__(256) void k_flow_upsample(const float2 *__restrict__ src, size_t src_stride, int sw, int sh,
                                                       float2 *__restrict__ dst, size_t dst_stride, int dw, int dh, float scale)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    dst[blockIdx.z * dst_stride + (size_t)y * dw + x] = flow_upsample_cell(src + blockIdx.z * src_stride, sw, sh, x, y, dw, dh, scale);
}

// What a refinement level needs before its Jacobi steps, in one launch: the derivatives of the level (k_hs_prepare on the
// luminance planes) and the coarser level's flow upsampled onto it (k_flow_upsample) -- both per cell, independent.
__global__ __launch_bounds__(256) void k_hs_level_setup(const float *__restrict__ l1, const float *__restrict__ l2, size_t lum_stride,
                                                        float *__restrict__ coef, size_t coef_stride, int w, int h,
                                                        const float2 *__restrict__ coarse, size_t coarse_stride, int cw, int ch,
                                                        float2 *__restrict__ flow, size_t flow_stride, float scale)
{
    const int x = blockIdx.x * kWave + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t z = blockIdx.z; // pair of the batch
    hs_prepare_cell(l1 + z * lum_stride, l2 + z * lum_stride, coef + z * coef_stride, x, y, w, h);
    flow[z * flow_stride + (size_t)y * w + x] = flow_upsample_cell(coarse + z * coarse_stride, cw, ch, x, y, w, h, scale);
}

} // namespace

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

// Batches: every launcher below takes `n` independent images / pairs on the grid's z axis, buffer b of item z at
// b + z * stride (strides in elements of the buffer's type; bytes for an RGBA8 input).  n = 1 ignores the strides.

// One fused pyramid level: `in` is RGBA8 (u8_input) or f32 RGBA; writes the level's luminance plane
// (w*h floats) and the f32 RGBA input of the next level; `next` may be null (last level).
#ifndef NUS_PYR_STREAM_WAVES
#define NUS_PYR_STREAM_WAVES 8192 // waves a launch of the streamed pyramid kernel aims for
#endif
#ifndef NUS_PYR_STREAM_MIN_ROWS
#define NUS_PYR_STREAM_MIN_ROWS 32 // shortest row block (4 halo rows are blurred horizontally per block)
#endif

// FAST pyramid level (k_pyramid_fast): luminance only.  `in`: RGBA8 frames (level 0; in_stride in bytes) or the previous level's
// quarter-size luminance output (floats; in_stride in floats); `next`: this level's output for the next one (floats), or null.
hipError_t launch_pyramid_level_fast(const void *in, bool u8_input, float *level_lum, float *next, uint32_t w, uint32_t h,
                                     hipStream_t stream, uint32_t n, size_t in_stride, size_t lum_stride, size_t next_stride)
{
#if defined(NUS_ABLATE_PYR_ALIGNED_STRIPS)
    const uint32_t strips = cdiv(w, 4 * kWave);
#else
    const uint32_t strips = cdiv(w, 4 * kWave - 8);
#endif
    const uint64_t columns = (uint64_t)strips * n;
    const uint32_t want = (uint32_t)std::min<uint64_t>((NUS_PYR_STREAM_WAVES + columns - 1) / columns,
                                                       std::max<uint32_t>(h / NUS_PYR_STREAM_MIN_ROWS, 1));
    const uint32_t rows_per_block = (cdiv(h, want) + 1) & ~1u; // even: a 2x2 block never straddles two row blocks
    const uint32_t row_blocks = cdiv(h, rows_per_block);
    const dim3 block(256), grid(cdiv(strips * row_blocks, 4), n);
    if (u8_input)
        hipLaunchKernelGGL(k_pyramid_fast<true>, grid, block, 0, stream, in, in_stride, level_lum, lum_stride, next, next_stride, (int)w,
                           (int)h, (int)strips, (int)row_blocks, (int)rows_per_block);
    else
        hipLaunchKernelGGL(k_pyramid_fast<false>, grid, block, 0, stream, in, in_stride, level_lum, lum_stride, next, next_stride, (int)w,
                           (int)h, (int)strips, (int)row_blocks, (int)rows_per_block);
    return hipGetLastError();
}

hipError_t launch_pyramid_level(const void *in, bool u8_input, float *level_lum, float *next, uint32_t w, uint32_t h,
                                hipStream_t stream, uint32_t n, size_t in_stride, size_t lum_stride, size_t next_stride, int kernel)
{
    if (kernel != kJacobiTiles) {
        const uint32_t strips = cdiv(w, 2 * kWave - 4);
        const uint64_t columns = (uint64_t)strips * n;
        const uint32_t want = (uint32_t)std::min<uint64_t>((NUS_PYR_STREAM_WAVES + columns - 1) / columns,
                                                           std::max<uint32_t>(h / NUS_PYR_STREAM_MIN_ROWS, 1));
        const uint32_t rows_per_block = (cdiv(h, want) + 1) & ~1u; // even: a 2x2 block never straddles two row blocks
        const uint32_t row_blocks = cdiv(h, rows_per_block);
        if (kernel == kJacobiStream || kernel == kJacobiStreamFast || columns * row_blocks >= 2048) {
            const dim3 block(256), grid(cdiv(strips * row_blocks, 4), n);
            if (u8_input)
                hipLaunchKernelGGL(k_pyramid_stream<true>, grid, block, 0, stream, in, in_stride, level_lum, lum_stride,
                                   reinterpret_cast<float4 *>(next), next_stride, (int)w, (int)h, (int)strips, (int)row_blocks,
                                   (int)rows_per_block);
            else
                hipLaunchKernelGGL(k_pyramid_stream<false>, grid, block, 0, stream, in, in_stride, level_lum, lum_stride,
                                   reinterpret_cast<float4 *>(next), next_stride, (int)w, (int)h, (int)strips, (int)row_blocks,
                                   (int)rows_per_block);
            return hipGetLastError();
        }
    }
    const dim3 block(kPyrThreads), grid(cdiv(w, kPyrTW), cdiv(h, kPyrTH), n);
    if (u8_input)
        hipLaunchKernelGGL(k_pyramid_level<true>, grid, block, 0, stream, in, in_stride, level_lum, lum_stride,
                           reinterpret_cast<float4 *>(next), next_stride, (int)w, (int)h);
    else
        hipLaunchKernelGGL(k_pyramid_level<false>, grid, block, 0, stream, in, in_stride, level_lum, lum_stride,
                           reinterpret_cast<float4 *>(next), next_stride, (int)w, (int)h);
    return hipGetLastError();
}

// coef: 3 floats (ix, iy, it) per cell -> w*h*12 bytes
hipError_t launch_hs_prepare(const float *i1, const float *i2, bool luminance_planes, float *coef, uint32_t w, uint32_t h,
                             hipStream_t stream, uint32_t n, size_t img_stride, size_t coef_stride)
{
    const dim3 block(kWave, 4), grid(cdiv(w, kWave), cdiv(h, 4), n);
    if (luminance_planes)
        hipLaunchKernelGGL(k_hs_prepare<float>, grid, block, 0, stream, i1, i2, img_stride, coef, coef_stride, (int)w, (int)h);
    else
        hipLaunchKernelGGL(k_hs_prepare<float4>, grid, block, 0, stream, reinterpret_cast<const float4 *>(i1),
                           reinterpret_cast<const float4 *>(i2), img_stride, coef, coef_stride, (int)w, (int)h);
    return hipGetLastError();
}

// prepare (luminance planes) + upsample of the coarser flow in one launch
hipError_t launch_hs_level_setup(const float *l1, const float *l2, float *coef, uint32_t w, uint32_t h, const float *coarse,
                                 uint32_t cw, uint32_t ch, float *flow, float scale, hipStream_t stream, uint32_t n,
                                 size_t lum_stride, size_t coef_stride, size_t coarse_stride, size_t flow_stride)
{
    const dim3 block(kWave, 4), grid(cdiv(w, kWave), cdiv(h, 4), n);
    hipLaunchKernelGGL(k_hs_level_setup, grid, block, 0, stream, l1, l2, lum_stride, coef, coef_stride, (int)w, (int)h,
                       reinterpret_cast<const float2 *>(coarse), coarse_stride, (int)cw, (int)ch,
                       reinterpret_cast<float2 *>(flow), flow_stride, scale);
    return hipGetLastError();
}

// `iterations` Jacobi steps from *flow_a (from zero flow without reading it if zero_start),
// ping-ponging with *flow_b; on return *flow_a holds the result (the pointers are swapped as
// needed; with final_out the last launch writes there -- item stride final_stride -- and *flow_a == final_out).
// Steps are split evenly over the launches.  Tile shape by the number of 32x32 tiles the whole batch has:
//   >= 1024 (1080p, or a batch of smaller levels): 32-wide tiles, 256 threads (7 cells per thread, 5 tiles per CU
//            by LDS), at most 5 steps per launch (beyond that the tile's registers and LDS cost a wave per SIMD);
//   >= 256  (one 960x540 level): 32-wide tiles with 1024 threads;
//   < 256   (one 480x270 level): 16-wide tiles, so that the grid still covers the 256 CUs, with 1024 threads of one
//            cell each: two tiles per CU leave the step's LDS round trip exposed unless the waves are many.
#ifndef NUS_HS_BIG_THREADS
#define NUS_HS_BIG_THREADS 256
#endif
#ifndef NUS_HS_MID_T
#define NUS_HS_MID_T 32
#endif
#ifndef NUS_HS_MID_THREADS
#define NUS_HS_MID_THREADS 1024
#endif
#ifndef NUS_HS_SMALL_T
#define NUS_HS_SMALL_T 16
#endif
#ifndef NUS_HS_SMALL_THREADS
#define NUS_HS_SMALL_THREADS 1024
#endif
#ifndef NUS_HS_STREAM_MAXK
#define NUS_HS_STREAM_MAXK 5 // steps per launch of the streamed kernel (registers: 8 per level + 5 per delay-line row)
#endif
#ifndef NUS_HS_FAST_MAXK_LONG
#define NUS_HS_FAST_MAXK_LONG 8 // ... of a level with >= 30 steps (50 steps: 7 launches instead of 10; -0.8 us per 1080p pair)
#endif
#ifndef NUS_HS_FAST_MAXK
#define NUS_HS_FAST_MAXK 5 // steps per launch of k_hs_stream_fast (registers: 4 per level + 5 per delay-line row)
#endif
#ifndef NUS_HS_STREAM_MIN_ROWS
#define NUS_HS_STREAM_MIN_ROWS 64 // shortest row block: 2K halo rows and the K passes of pipeline fill are paid per block
#endif
#ifndef NUS_HS_STREAM_MIN_WAVES
#define NUS_HS_STREAM_MIN_WAVES 1024 // below this the LDS tiles fill the chip better
#endif

// The streamed kernel's launch shape for a level: strips of 64 - 2K columns, cut into row blocks.  The kernel is bound
// by instruction issue and all its waves take the same time, so a launch costs (waves per SIMD, rounded up) x (passes
// per wave = rows of a block + 2K halo rows + K passes of fill); the number of row blocks minimises that among the
// shapes with at least three waves per SIMD (fewer do not cover each other's latencies).
// row_blocks == 0: not enough independent strips, use the LDS tiles.
struct HsStreamShape {
    uint32_t strips, row_blocks, rows_per_block;
};
static HsStreamShape hs_stream_shape(uint32_t w, uint32_t h, uint32_t n, uint32_t k, bool force)
{
    constexpr uint64_t kSimds = 1024; // 256 CUs x 4
    HsStreamShape s{cdiv(w, kWave - 2 * k), 1, h};
    const uint64_t columns = (uint64_t)s.strips * n;
    const uint32_t max_blocks = std::max<uint32_t>(h / NUS_HS_STREAM_MIN_ROWS, 1);
    uint64_t best = ~0ull;
    for (uint32_t rb = 1; rb <= max_blocks; ++rb) {
        const uint32_t rows = cdiv(h, rb);
        if (cdiv(h, rows) != rb) continue; // the same blocks as a smaller count
        const uint64_t waves = columns * rb;
        if (waves < 3 * kSimds && rb != max_blocks) continue;
        const uint64_t cost = ((waves + kSimds - 1) / kSimds) * (rows + 3 * k);
        if (cost < best) best = cost, s.row_blocks = rb, s.rows_per_block = rows;
    }
    if (!force && columns * s.row_blocks < NUS_HS_STREAM_MIN_WAVES) s.row_blocks = 0;
    return s;
}

// true: launch_hs_iterate will run the streamed kernel for this level -- which can take the derivatives from the
// luminance planes (lum1), so the caller need not have k_hs_prepare's coefficient planes written
bool hs_iterate_streams(uint32_t w, uint32_t h, uint32_t n, int kernel)
{
    return kernel != kJacobiTiles &&
           hs_stream_shape(w, h, n, NUS_HS_STREAM_MAXK, kernel == kJacobiStream || kernel == kJacobiStreamFast).row_blocks != 0;
}

hipError_t launch_hs_iterate(const float *coef, float lambda, float **flow_a, float **flow_b, uint32_t w, uint32_t h,
                             uint32_t iterations, bool zero_start, float *final_out, hipStream_t stream, uint32_t n,
                             size_t coef_stride, size_t flow_stride, size_t final_stride, int kernel, const float *lum1,
                             size_t lum_stride, const float *coarse, uint32_t cw, uint32_t ch, float coarse_scale, size_t coarse_stride,
                             const HsWarp *warp, bool *warped, bool *wrote_half)
{
    if (kernel == kJacobiStreamFast) { // FAST arithmetic (k_hs_stream_fast): always streamed, always from the luminance planes
        if (lum1 == nullptr) return hipErrorInvalidValue;
        // one launch of k steps over pairs [0, m) of the given bases
        bool did_warp = false, did_half = false;
        auto launch_one = [&](uint32_t k, bool ups, const float *lum, const float2 *fi, float2 *fo, size_t out_stride, const HsCoarse &hc,
                              uint32_t m, const HsWarp *wp, bool half_out, bool half_in) -> hipError_t {
            HsWarp tail; // what the plain instantiations see of it: only the format of the flow they store (and load)
            tail.out_half = half_out ? 1u : 0u;
            tail.in_half = half_in ? 1u : 0u;
            HsWarp wfull = wp ? *wp : HsWarp{};
            wfull.out_half = tail.out_half;
            wfull.in_half = tail.in_half;
            if (wp) wp = &wfull;
            const HsStreamShape sh = hs_stream_shape(w, h, m, k, true);
            const dim3 block(256), grid(cdiv(sh.strips * sh.row_blocks, 4), m);
#define NUS_HSF_L(KK, UU, RR)                                                                                                        \
    hipLaunchKernelGGL((k_hs_stream_fast<KK, UU, RR>), grid, block, 0, stream, lum, lum_stride, lambda, fi, flow_stride, fo, out_stride, \
                       (int)w, (int)h, (int)sh.strips, (int)sh.row_blocks, (int)sh.rows_per_block, hc, tail)
#define NUS_HSF_W(KK)                                                                                                                \
    hipLaunchKernelGGL((k_hs_stream_fast<KK, false, true, true>), grid, block, 0, stream, lum, lum_stride, lambda, fi, flow_stride, fo, \
                       out_stride, (int)w, (int)h, (int)sh.strips, (int)sh.row_blocks, (int)sh.rows_per_block, hc, *wp)
#define NUS_HSF(KK)                                                                                                                  \
    case KK:                                                                                                                         \
        if constexpr (NUS_HS_FAST_RING != 0 && KK <= 5) {                                                                            \
            if (wp != nullptr && !shifting && !ups) {                                                                                \
                NUS_HSF_W(KK);                                                                                                       \
                did_warp = true;                                                                                                     \
                break;                                                                                                               \
            }                                                                                                                        \
        }                                                                                                                            \
        if constexpr (NUS_HS_FAST_RING != 0 && KK <= NUS_HS_FAST_RING_MAXK) {                                                                            \
            if (!shifting) {                                                                                                         \
                if (ups) NUS_HSF_L(KK, true, true); else NUS_HSF_L(KK, false, true);                                                 \
                break;                                                                                                               \
            }                                                                                                                        \
        }                                                                                                                            \
        if (ups) NUS_HSF_L(KK, true, false); else NUS_HSF_L(KK, false, false);                                                       \
        break;
            // (test hook: NUS_HS_FAST_SHIFT=1 in the environment runs the shifting form of the pass where the ring form is the
            // product's -- tests/test_flow.py compares the two bit for bit)
            const char *shift_env = getenv("NUS_HS_FAST_SHIFT");
            const bool shifting = shift_env != nullptr && shift_env[0] == '1';
            switch (k) {
                NUS_HSF(1) NUS_HSF(2) NUS_HSF(3) NUS_HSF(4) NUS_HSF(5)
#if NUS_HS_FAST_MAXK > 5 || NUS_HS_FAST_MAXK_LONG > 5
                NUS_HSF(6) NUS_HSF(7) NUS_HSF(8) NUS_HSF(9) NUS_HSF(10)
#endif
            }
#undef NUS_HSF
#undef NUS_HSF_L
#undef NUS_HSF_W
            return hipGetLastError();
        };
        // a level with many steps (the coarsest: 50) takes more of them per launch: its launches are short and memory-bound, and
        // fewer of them move fewer bytes; the levels with 10 steps stay at 5 (ten per launch costs two waves per SIMD)
        const uint32_t maxk = iterations >= 30 ? NUS_HS_FAST_MAXK_LONG : NUS_HS_FAST_MAXK;
        uint32_t launches = (iterations + maxk - 1) / maxk;
        const char *hb_env = getenv("NUS_HS_L0_HALF_BETWEEN"); // dev switch: see HsWarp::in_half
        const bool half_between = hb_env != nullptr && hb_env[0] == '1' && warp != nullptr && launches == 2 && !zero_start;
        bool prev_stored_half = false;
        // (Measured and dropped, round 4: the finest level in sub-chunks of 4-50 pairs, its two launches back to back per sub-chunk so
        // that the second finds the first's output and the luminance planes in the 256-MiB Infinity Cache -- identical flows,
        // 61 / 58 / 55 / 54 / 53 / 52 / 50 us per pair at 4 / 6 / 8 / 12 / 16 / 25 / 50 pairs against 50.5 for the whole chunk: launches
        // of a few thousand waves lose more than the cache gives.)
        while (iterations > 0) {
            const uint32_t k = (iterations + launches - 1) / launches; // even split, 1..maxk steps per launch
            size_t out_stride = flow_stride;
            if (final_out && launches == 1) { // the last launch writes the caller's buffer
                *flow_b = final_out;
                out_stride = final_stride;
            }
            auto fi = zero_start ? nullptr : reinterpret_cast<const float2 *>(*flow_a);
            auto fo = reinterpret_cast<float2 *>(*flow_b);
            zero_start = false;
            const bool ups = coarse != nullptr; // the first launch of a level takes the coarser level's flow, upsampled as it loads it
            const HsCoarse hc{reinterpret_cast<const float2 *>(coarse), coarse_stride, (int)cw, (int)ch, coarse_scale};
            coarse = nullptr;
            // the last launch of the level warps with the flow it finishes, where it can (see HsWarp); then the flow is stored only
            // for a caller who asked for it
            // OFF unless NUS_HS_FUSED_WARP=1 is in the environment: built and measured in round 5 -- identical bytes, but the fused launch
            // takes 23 us per 1080p pair where the plain launch and the warp kernel take 12.2 + 7.9: the Jacobi pass has one pixel per
            // lane, so the warp's gathers are 8-byte loads at a 4-byte lane stride (every texel pair fetched twice) and its ~140
            // instructions per pixel run at four waves per SIMD (119 VGPRs) next to a pass that was at the memory rate, not below it
            // (motion step 21.0 against 20.1 ms per 300 units).  The warp kernel's 2 x 2 pixels per thread is the better shape.
            const char *fuse_env = getenv("NUS_HS_FUSED_WARP");
            const bool can_warp = fuse_env != nullptr && fuse_env[0] == '1' &&
                                  warp != nullptr && warp->frames != nullptr && warp->mid != nullptr && launches == 1 && !ups && k <= 5 &&
                                  NUS_HS_FAST_RING != 0 && w >= 2 && h >= 2 && (uint64_t)w * h * 4 < (1ull << 32) &&
                                  (uint64_t)w * 4 < (1u << 24) && h < (1u << 24);
            if (can_warp && final_out == nullptr) fo = nullptr;
            const bool half_out = warp != nullptr && warp->out_half != 0 && launches == 1; // the level's final flow as Rg16Float
            if (half_out) did_half = true;
            const bool mid_half = half_between && launches == 2; // the first of the two launches stores halves for the second
            hipError_t e = launch_one(k, ups, lum1, fi, fo, out_stride, hc, n, can_warp ? warp : nullptr, half_out || mid_half, prev_stored_half);
            prev_stored_half = mid_half;
            if (e != hipSuccess) return e;
            iterations -= k;
            --launches;
            float *t = *flow_a;
            *flow_a = *flow_b;
            *flow_b = t;
        }
        if (warped) *warped = did_warp;
        if (wrote_half) *wrote_half = did_half;
        return hipSuccess;
    }
    if (warped) *warped = false;
    if (wrote_half) *wrote_half = false;
    if (hs_iterate_streams(w, h, n, kernel)) {
        if (lum1) coef = lum1, coef_stride = lum_stride; // the kernel takes the derivatives from the planes themselves
        uint32_t launches = (iterations + NUS_HS_STREAM_MAXK - 1) / NUS_HS_STREAM_MAXK;
        while (iterations > 0) {
            const uint32_t k = (iterations + launches - 1) / launches; // even split, 1..MAXK steps per launch
            size_t out_stride = flow_stride;
            if (final_out && launches == 1) { // the last launch writes the caller's buffer
                *flow_b = final_out;
                out_stride = final_stride;
            }
            auto fi = zero_start ? nullptr : reinterpret_cast<const float2 *>(*flow_a);
            auto fo = reinterpret_cast<float2 *>(*flow_b);
            zero_start = false;
            const HsStreamShape sh = hs_stream_shape(w, h, n, k, true);
            // the first launch of a level that continues a coarser one takes that level's flow, upsampled as it is loaded
            const bool ups = coarse != nullptr && lum1 != nullptr;
            const HsCoarse hc{reinterpret_cast<const float2 *>(coarse), coarse_stride, (int)cw, (int)ch, coarse_scale};
            coarse = nullptr;
            const dim3 block(256), grid(cdiv(sh.strips * sh.row_blocks, 4), n);
#define NUS_HSS_L(KK, LL, UU)                                                                                               \
    hipLaunchKernelGGL((k_hs_stream<KK, LL, UU>), grid, block, 0, stream, coef, coef_stride, lambda, fi, flow_stride, fo, out_stride, \
                       (int)w, (int)h, (int)sh.strips, (int)sh.row_blocks, (int)sh.rows_per_block, hc)
#define NUS_HSS(KK)                                                                                                             \
    case KK:                                                                                                                    \
        if (lum1 && ups)                                                                                                        \
            NUS_HSS_L(KK, true, true);                                                                                          \
        else if (lum1)                                                                                                          \
            NUS_HSS_L(KK, true, false);                                                                                         \
        else                                                                                                                    \
            NUS_HSS_L(KK, false, false);                                                                                        \
        break;
            switch (k) {
                NUS_HSS(1) NUS_HSS(2) NUS_HSS(3) NUS_HSS(4) NUS_HSS(5)
#if NUS_HS_STREAM_MAXK > 5
                NUS_HSS(6) NUS_HSS(7) NUS_HSS(8)
#endif
#if NUS_HS_STREAM_MAXK > 8
                NUS_HSS(9) NUS_HSS(10)
#endif
            }
#undef NUS_HSS
#undef NUS_HSS_L
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
    const uint64_t tiles32 = (uint64_t)cdiv(w, 32) * cdiv(h, 32) * n;
    const int cls = tiles32 >= 1024 ? 2 : (tiles32 >= 256 ? 1 : 0);
    const uint32_t T = cls == 2 ? 32 : (cls == 1 ? NUS_HS_MID_T : NUS_HS_SMALL_T);
    const uint32_t NT = cls == 2 ? NUS_HS_BIG_THREADS : (cls == 1 ? NUS_HS_MID_THREADS : NUS_HS_SMALL_THREADS);
    const dim3 block(NT), grid(cdiv(w, T), cdiv(h, T), n);
    const uint32_t maxk = cls == 2 ? 5 : 8;
    uint32_t launches = (iterations + maxk - 1) / maxk;
    while (iterations > 0) {
        const uint32_t k = (iterations + launches - 1) / launches; // even split, 1..maxk steps per launch
        size_t out_stride = flow_stride;
        if (final_out && launches == 1) { // the last launch writes the caller's buffer
            *flow_b = final_out;
            out_stride = final_stride;
        }
        auto fi = zero_start ? nullptr : reinterpret_cast<const float2 *>(*flow_a);
        auto fo = reinterpret_cast<float2 *>(*flow_b);
        zero_start = false;
#define NUS_HS_L(TT, KK, TH)                                                                                                  \
    hipLaunchKernelGGL((k_hs_tiled<TT, KK, TH>), grid, block, 0, stream, coef, coef_stride, lambda, fi, flow_stride, fo, out_stride, \
                       (int)w, (int)h)
#define NUS_HS(KK)                                                    \
    case KK:                                                          \
        if (cls == 2) NUS_HS_L(32, KK, NUS_HS_BIG_THREADS);           \
        else if (cls == 1) NUS_HS_L(NUS_HS_MID_T, KK, NUS_HS_MID_THREADS); \
        else NUS_HS_L(NUS_HS_SMALL_T, KK, NUS_HS_SMALL_THREADS);      \
        break;
        switch (k) {
            NUS_HS(1) NUS_HS(2) NUS_HS(3) NUS_HS(4) NUS_HS(5) NUS_HS(6) NUS_HS(7) NUS_HS(8)
        }
#undef NUS_HS
#undef NUS_HS_L
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

namespace {
__global__ __launch_bounds__(256) void k_flow_to_half(const float2 *__restrict__ src, __half2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float2 f = src[i];
        dst[i] = __floats2half2_rn(f.x, f.y);
    }
}
} // namespace

hipError_t launch_flow_to_half(const float *src, void *dst, size_t n_cells, hipStream_t stream)
{
    if (n_cells == 0) return hipSuccess;
    hipLaunchKernelGGL(k_flow_to_half, dim3((uint32_t)((n_cells + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const float2 *>(src),
                       static_cast<__half2 *>(dst), n_cells);
    return hipGetLastError();
}

hipError_t launch_flow_upsample(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh,
                                float scale, hipStream_t stream, uint32_t n, size_t src_stride, size_t dst_stride)
{
    const dim3 block(kWave, 4), grid(cdiv(dw, kWave), cdiv(dh, 4), n);
    hipLaunchKernelGGL(k_flow_upsample, grid, block, 0, stream, reinterpret_cast<const float2 *>(src), src_stride, (int)sw, (int)sh,
                       reinterpret_cast<float2 *>(dst), dst_stride, (int)dw, (int)dh, scale);
    return hipGetLastError();
}

} // namespace nus

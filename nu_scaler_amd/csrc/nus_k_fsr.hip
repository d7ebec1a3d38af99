// nus_k_fsr.hip -- FSR1-style EASU + RCAS.
#include "nus_device.hpp"

namespace nus {

namespace {

// ---------------------------------------------------------------------------------
// FSR1-style EASU + RCAS: nu_scaler_core/src/upscale/fsr.rs:24-260
// ---------------------------------------------------------------------------------
// One kernel, three modes.  Stage 1 fills an LDS tile of packed RGBA8 pixels -- EASU evaluations
// (modes Easu, Fused) or plain loads (mode Rcas) -- with a 1-pixel halo when RCAS follows; stage 2
// writes the tile out, through the 5-tap RCAS when asked.  The fused mode therefore never writes the
// EASU image to HBM (the shader pair round-trips it as RGBA8, which the LDS tile reproduces exactly:
// same truncating pack between the passes).  Expression order follows the shaders; no contraction.
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

// EASU direction weight wx of FsrDirA + fsr.rs:131-133, from the four unpacked neighbours.  The shader's five divisions in three
// IEEE ones: x / 3.0 through the constant's reciprocal and dxr / len, dyr / len through one reciprocal of len (div_by_recip: the
// correctly rounded quotient; a len whose mantissa is all ones takes the plain divisions) -- the same bits as before.
__device__ __forceinline__ float fsr_dir_wx(const float3 up, const float3 dn, const float3 lf, const float3 rt)
{
    const float third = 1.0f / 3.0f;
    const float vgx = div_by_recip(fabsf(up.x - dn.x) + fabsf(up.y - dn.y) + fabsf(up.z - dn.z), 3.0f, third);
    const float vgy = div_by_recip(fabsf(lf.x - rt.x) + fabsf(lf.y - rt.y) + fabsf(lf.z - rt.z), 3.0f, third);
    const float dxr = vgx + 0.0001f, dyr = vgy + 0.0001f;
    const float len = sqrtf(dxr * dxr + dyr * dyr);
    float dirx, diry;
    if ((__float_as_uint(len) & 0x007fffffu) == 0x007fffffu) {
        dirx = dxr / len, diry = dyr / len;
    } else {
        const float z = 1.0f / len;
        dirx = div_by_recip(dxr, len, z), diry = div_by_recip(dyr, len, z);
    }
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

// ---- FAST arithmetic (option "fsr_fast"; round 5) --------------------------------------------------------------------------
// The exact kernel reproduces, operation for operation, two shaders the reference never dispatches (its FsrUpscaler returns "not
// implemented", fsr.rs:316-332), at ~450 f32 instructions per output pixel of the fused pair.  FAST keeps the algorithm and the
// decisions and drops the operation order where no decision depends on it:
//   * a tap's distance d = |(x - fx) wx + (y - fy) wy| is computed exactly as the shader does (same products, same sum: the
//     comparisons d <= 1, d <= 2 of FsrCubic -- the second one is a DISCONTINUITY of the weight, 1 -> 0 -- take the shader's side for
//     every tap), scaled by 128 through wx, wy (exact: a power of two);
//   * FsrCubic(d) from a table in LDS: 128 cells per unit of d, each cell (value at its left end, difference to its right end) inside
//     ONE polynomial piece (the pieces meet at d = 1 = cell 128 and d = 2 = cell 256), linear interpolation with the cell's own
//     fraction: |error| <= h^2 / 8 max|f''| = 5.3e-5 on weights of order 1; cell 256 (d = 2 exactly: weight 1) carries a slope that sends
//     any d beyond it below zero, and the weight is max(., 0): 6 instructions + one ds_read_b64 where the two cubics, two compares and
//     two selects take 13;
//   * the 16-tap sums as FMAs, the division by the weight sum as v_rcp_f32 + one Newton step.
// The direction weight of a texel (once per INPUT texel in the LDS prepass) and RCAS stay in the shader's own arithmetic.
// Contract (tests/test_fsr1.py): EASU within 1 LSB of orc_fsr_easu (the truncating pack turns a 1e-6 difference into a count at
// integer boundaries -- the exact pair itself returns {76, 77} for a flat 77); the fused pair is bit for bit orc_fsr_rcas applied to
// the FAST EASU image (RCAS amplifies a count of its centre tap by 1 + 4 sharpness, so "within 1 LSB of orc_fsr1" is not a contract
// any reordering of EASU can meet; EXACT remains the default and the verification mode).
constexpr int kFsrLutScale = 128, kFsrLutCells = 3 * kFsrLutScale + 2; // d <= 3 (fx = fy = 0, tap (3, 3)): cells 0 .. 384 are reachable

__device__ __forceinline__ float fsr_cubic_piece(float d, bool near_piece) // the shader's polynomials, plainly
{
    const float d2 = d * d, d3 = d * d2;
    return near_piece ? 2.0f - 1.5f * d - 0.5f * d3 + d2 : 0.0f - 0.5f * d + 2.5f * d2 - d3;
}

__device__ __forceinline__ void fsr_build_lut(float2 *lut, int tid)
{
    for (int i = tid; i < kFsrLutCells; i += 256) {
        const float a = (float)i / (float)kFsrLutScale, b = (float)(i + 1) / (float)kFsrLutScale;
        float2 e = make_float2(0.0f, 0.0f);
        if (i < 2 * kFsrLutScale) {
            const bool near_piece = i < kFsrLutScale;
            const float va = fsr_cubic_piece(a, near_piece);
            e = make_float2(va, fsr_cubic_piece(b, near_piece) - va);
        } else if (i == 2 * kFsrLutScale) {
            e = make_float2(fsr_cubic_piece(2.0f, false), -1.0e9f); // d == 2 exactly keeps its weight; anything beyond goes negative
        }
        // HALF the weight (exact: a power of two; the sums and the quotient scale with it, bit for bit): FsrCubic runs from 0 to 2, half
        // of it fits the [0, 1] clamp that the interpolating FMA applies for free (fsr_lut_weight)
        lut[i] = make_float2(0.5f * e.x, 0.5f * e.y);
    }
}

// half the FsrCubic weight of a tap at scaled distance ds = 128 d >= 0: cell = floor(ds), linear inside the cell, clamped to [0, 1]
// (the clamp is the FMA's output modifier; it is what makes everything beyond d = 2 zero).  v_fract_f32 + v_cvt_u32_f32 + the address
// shift + ds_read_b64 + v_fma_f32 clamp.
__device__ __forceinline__ float fsr_lut_weight(const float2 *lut, float ds)
{
    const float2 e = lut[(uint32_t)ds];
    return __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_amdgcn_fractf(ds), e.y, e.x), 0.0f, 1.0f);
}

template <typename TAP>
__device__ __forceinline__ uint32_t fsr_easu_taps_fast(TAP &&tap, const float2 *lut, float wx, float fx, float fy, float sharp)
{
    const float wy = 1.0f - wx;
    const float wxs = wx * (float)kFsrLutScale, wys = wy * (float)kFsrLutScale;
    float X[4], Y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        X[k] = ((float)k - fx) * wxs; // == (((float)k - fx) * wx) * 128: the shader's product, scaled exactly
        Y[k] = ((float)k - fy) * wys;
    }
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float4 c = tap(x, y);
            const float ds = fabsf(X[x] + Y[y]); // 128 d, d as the shader rounds it
            const float wgt = fsr_lut_weight(lut, ds);
            sr = __builtin_fmaf(c.x, wgt, sr);
            sg = __builtin_fmaf(c.y, wgt, sg);
            sb = __builtin_fmaf(c.z, wgt, sb);
            sw += wgt;
        }
    }
    const float den = fmaxf(sw, 0.5f * 0.0001f); // (sw is half the shader's sum)
    float z = __builtin_amdgcn_rcpf(den);
    z = z * __builtin_fmaf(-den, z, 2.0f); // one Newton step: relative error ~1e-7
    sr *= z, sg *= z, sb *= z;
    if (sharp > 0.001f) {
        const float4 ctr = tap(1, 1);
        const float ns = 1.0f - sharp;
        sr = __builtin_fmaf(sr, ns, ctr.x * sharp);
        sg = __builtin_fmaf(sg, ns, ctr.y * sharp);
        sb = __builtin_fmaf(sb, ns, ctr.z * sharp);
    }
    return fsr_pack(sr, sg, sb);
}

// FAST at exactly x2: output pixels (2k + a, 2m + b), a, b in {0, 1}, all have (ix, iy) = (k, m) -- one 4x4 block of taps and one
// direction weight for the four of them -- and fx, fy in {0.25, 0.75} exactly: the taps are read from LDS once per quad (a quarter of
// the kernel's LDS traffic, which bounds it next to instruction issue) and the eight scaled offsets per axis are shared.  Per pixel
// the operations are those of fsr_easu_taps_fast, in its order: the same bytes (tests/test_fsr1.py runs both).
template <typename TAP>
__device__ __forceinline__ void fsr_easu_quad_fast(TAP &&tap, const float2 *lut, float wx, float sharp, uint32_t (&out)[4])
{
    const float wy = 1.0f - wx;
    const float wxs = wx * (float)kFsrLutScale, wys = wy * (float)kFsrLutScale;
    float4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = tap(i & 3, i >> 2);
    float X[2][4], Y[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = a ? 0.75f : 0.25f;
            X[a][k] = ((float)k - f) * wxs;
            Y[a][k] = ((float)k - f) * wys;
        }
    const float ns = 1.0f - sharp;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    // (tap (3, 3) never counts at x2 -- 2.75 or 2.25 texels away on both axes --, but leaving it out breaks the pairing of
                    // the taps' packed adds: 38.3 -> 41.1 us, measured)
                    const float ds = fabsf(X[a][x] + Y[b][y]);
                    const float wgt = fsr_lut_weight(lut, ds);
                    sr = __builtin_fmaf(c[y * 4 + x].x, wgt, sr);
                    sg = __builtin_fmaf(c[y * 4 + x].y, wgt, sg);
                    sb = __builtin_fmaf(c[y * 4 + x].z, wgt, sb);
                    sw += wgt;
                }
            const float den = fmaxf(sw, 0.5f * 0.0001f); // (sw is half the shader's sum)
            float z = __builtin_amdgcn_rcpf(den);
            z = z * __builtin_fmaf(-den, z, 2.0f);
            sr *= z, sg *= z, sb *= z;
            if (sharp > 0.001f) {
                sr = __builtin_fmaf(sr, ns, c[5].x * sharp);
                sg = __builtin_fmaf(sg, ns, c[5].y * sharp);
                sb = __builtin_fmaf(sb, ns, c[5].z * sharp);
            }
            out[b * 2 + a] = fsr_pack(sr, sg, sb);
        }
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
template <FsrMode MODE, bool VEC, bool SRC_LDS, bool FAST = false, bool QUAD = false>
__global__ __launch_bounds__(256) void k_fsr1(const FsrArgs A, const int src_cap)
{
    static_assert(!FAST || (SRC_LDS && MODE != FsrMode::Rcas), "FAST: EASU out of the LDS source tile");
    static_assert(!QUAD || FAST, "QUAD (exactly x2, host-checked): a variant of FAST");
    __shared__ float2 lut[FAST ? kFsrLutCells : 1];
    if constexpr (FAST) fsr_build_lut(lut, threadIdx.x); // (visible after the source tile's barriers)
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
    if constexpr (QUAD) {
        // one thread per 2x2 quad of output pixels (see fsr_easu_quad_fast); quads are aligned to even output coordinates, so with
        // the halo of the fused mode the quad grid starts two pixels before the tile and covers one row / column more than it on
        // each side (9 % of the quads' pixels are not stored).  Quads outside the image are skipped: the halo cells they would
        // have filled hold the clamped pixel's value, copied below.
        constexpr int Q0 = HALO ? 2 : 0, NQX = (kFsrTW + 2 * Q0) / 2, NQY = (kFsrTH + 2 * Q0) / 2;
        constexpr int ROUNDS = (NQX * NQY + 255) / 256; // quads per thread (2 for EASU alone, 3 with the halo)
        // the packed pixels of a thread's quads stay in registers until all of them are computed: unpacking them for RCAS next to
        // the 16 taps and the tables of the quad still being summed cost 234 registers (two waves per SIMD)
        uint32_t keep[ROUNDS][4];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int q = tid + 256 * r;
            const int qy = q / NQX, qx = q - qy * NQX;
            const int k = (x0 - Q0) / 2 + qx, m = (y0 - Q0) / 2 + qy; // x0, y0 are even
#pragma unroll
            for (int j = 0; j < 4; ++j) keep[r][j] = 0u;
            if (q >= NQX * NQY || k < 0 || k >= A.iw || m < 0 || m >= A.ih) continue;
            const float4 *t = src + (m - 1 - fy0) * fw + (k - 1 - fx0);
            fsr_easu_quad_fast([&](int x, int y) { return t[y * fw + x]; }, lut, t[fw + 1].w, A.easu_sharp, keep[r]);
            if constexpr (MODE == FsrMode::Easu) { // no halo: the quad is four cells of the tile (87 registers this way, 102 held back)
                tile[(2 * qy) * LW + 2 * qx] = keep[r][0], tile[(2 * qy) * LW + 2 * qx + 1] = keep[r][1];
                tile[(2 * qy + 1) * LW + 2 * qx] = keep[r][2], tile[(2 * qy + 1) * LW + 2 * qx + 1] = keep[r][3];
            } else {
                asm volatile("" : "+v"(keep[r][0]), "+v"(keep[r][1]), "+v"(keep[r][2]), "+v"(keep[r][3])); // (keeps the rounds apart)
            }
        }
        if constexpr (MODE != FsrMode::Easu)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int q = tid + 256 * r;
            const int qy = q / NQX, qx = q - qy * NQX;
            const int k = (x0 - Q0) / 2 + qx, m = (y0 - Q0) / 2 + qy;
            if (q >= NQX * NQY || k < 0 || k >= A.iw || m < 0 || m >= A.ih) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int lx = 2 * qx + a - (Q0 - HALO), ly = 2 * qy + b - (Q0 - HALO);
                    if (lx < 0 || lx >= LW || ly < 0 || ly >= LH) continue;
                    if constexpr (MODE == FsrMode::Easu)
                        tile[ly * LW + lx] = keep[r][b * 2 + a];
                    else
                        tile[ly * LW + lx] = fsr_rcas_tap(keep[r][b * 2 + a]);
                }
        }
        if (HALO && (x0 == 0 || y0 == 0 || x0 + kFsrTW >= A.ow || y0 + kFsrTH >= A.oh)) { // (block-uniform) a tile at the image's border
            __syncthreads();
            for (int i = tid; i < LW * LH; i += 256) {
                const int ly = i / LW, lx = i - ly * LW;
                const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
                const int cgx = clampi(gx, 0, A.ow - 1), cgy = clampi(gy, 0, A.oh - 1);
                const int cl = (cgy - y0 + HALO) * LW + (cgx - x0 + HALO);
                // (cells further out than the tile can reach are never read; the clamped cell is an in-image cell of this tile
                // whenever the cell itself is one RCAS reads)
                if ((cgx != gx || cgy != gy) && cgx - x0 + HALO < LW && cgy - y0 + HALO < LH) tile[i] = tile[cl];
            }
        }
    } else
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
            if constexpr (FAST)
                p = fsr_easu_taps_fast([&](int x, int y) { return t[y * fw + x]; }, lut, t[fw + 1].w, cx - floorf(cx),
                                       cy - floorf(cy), A.easu_sharp);
            else
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

// ---- RCAS as a row walker (round 5) ------------------------------------------------------------------------------------------
// The tile form of RCAS above (mode Rcas of k_fsr1) unpacks a 66 x 34 tile into LDS, waits at a barrier and filters out of it: its
// three phases -- loads, the LDS round trip, arithmetic + stores -- add up instead of overlapping (33 us per 4K frame for 65
// instructions and 66 MB per frame).  Here a wave owns a strip of 248 columns (lane L: columns 4 (L - 1) .. 4 (L - 1) + 3 of the
// strip; lanes 0 and 63 are the halo, as in the x2 resize kernel) and walks down its rows keeping THREE unpacked rows (rgb / 255 and
// luma per pixel: the shader's own unpack, fsr_rcas_tap) in registers: top / centre / bottom taps come from the window, left / right
// from the lane's own pixels or the neighbouring lane's (DPP), every input pixel is loaded and unpacked once per strip, no LDS, no
// barrier.  Coordinates outside the image are clamped as the shader's fetches are (the loads clamp: a clamped row / column is the
// border pixel again).  Same expressions as fsr_rcas_px: identical bits (tests/test_fsr1.py).
constexpr int kRcasStripCols = 248;

struct RcasWalkArgs {
    const uint32_t *in;
    uint32_t *out;
    int w, h;
    size_t px; // pixels per frame
    int strips, row_blocks, rows_per_block;
    float sharp;
    uint32_t sel;
};

struct RcasRow {
    float4 p[4]; // this lane's four pixels, unpacked (rgb in xyz, luma in w)
};

__device__ __forceinline__ float4 dpp_up4(const float4 v) { return make_float4(wave_up(v.x), wave_up(v.y), wave_up(v.z), wave_up(v.w)); }
__device__ __forceinline__ float4 dpp_down4(const float4 v) { return make_float4(wave_down(v.x), wave_down(v.y), wave_down(v.z), wave_down(v.w)); }

template <bool VEC>
__global__ __launch_bounds__(256) void k_fsr_rcas_walk(const RcasWalkArgs A)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= A.strips * A.row_blocks) return;
    const int rb = g / A.strips, strip = g - rb * A.strips;
    const uint32_t *__restrict__ in = A.in + (size_t)blockIdx.y * A.px;
    uint32_t *__restrict__ out = A.out + (size_t)blockIdx.y * A.px;
    const int c0 = strip * kRcasStripCols - 4 + lane * 4; // first column of this lane
    const int y0 = rb * A.rows_per_block, y1 = min(y0 + A.rows_per_block, A.h);
    const bool writer = lane >= 1 && lane <= kRcasStripCols / 4 && c0 < A.w;
    // columns of the lane's four pixels, clamped into the image (a clamped column IS the border pixel, as the shader fetches it)
    int cc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cc[k] = clampi(c0 + k, 0, A.w - 1);
    const bool vec_ok = VEC && c0 >= 0 && c0 + 4 <= A.w; // one 16-byte load
    struct RawRow {
        uint32_t px[4];
    };
    auto load_raw = [&](int y) -> RawRow {
        const uint32_t *row = in + (size_t)clampi(y, 0, A.h - 1) * A.w;
        RawRow r;
        if (vec_ok) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + c0);
            r.px[0] = v.x, r.px[1] = v.y, r.px[2] = v.z, r.px[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) r.px[k] = row[cc[k]];
        }
        return r;
    };
    auto unpack = [&](const RawRow &raw, RcasRow &r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r.p[k] = fsr_rcas_tap(swz(raw.px[k], A.sel));
    };
    // The lane's left / right neighbours in a row are its own pixels or the neighbouring lane's last / first.  The image's first and
    // last column need nothing special: positions outside the image were LOADED clamped, so the position left of column 0 (the halo
    // lane's last pixel) holds column 0 itself and the position right of column w - 1 holds column w - 1 -- the shader's clamp.
    auto filter_row = [&](int y, const RcasRow &top, const RcasRow &ctr, const RcasRow &bot) {
        const float4 from_left = dpp_up4(ctr.p[3]), from_right = dpp_down4(ctr.p[0]);
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 l = k == 0 ? from_left : ctr.p[k - 1];
            const float4 r = k == 3 ? from_right : ctr.p[k + 1];
            o[k] = fsr_rcas_px(ctr.p[k], top.p[k], bot.p[k], l, r, A.sharp);
        }
        if (writer) {
            uint32_t *dst = out + (size_t)y * A.w + c0;
            if (VEC && c0 + 4 <= A.w) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + k < A.w) dst[k] = o[k];
            }
        }
    };
    // three rows in registers, their roles (top / centre / bottom) rotating with the row: three rows of the walk unrolled, the incoming
    // row -- requested before the row's arithmetic, in flight during it -- unpacked straight into the registers of the row that has
    // just left the window (round 5, first form: `top = ctr, ctr = bot, bot = nxt`, 36 register moves per row)
    RcasRow r0, r1, r2;
    unpack(load_raw(y0 - 1), r0);
    unpack(load_raw(y0), r1);
    unpack(load_raw(y0 + 1), r2);
    for (int y = y0; y < y1; y += 3) {
        RawRow nxt = load_raw(y + 2);
        filter_row(y, r0, r1, r2);
        unpack(nxt, r0);
        if (y + 1 >= y1) break;
        nxt = load_raw(y + 3);
        filter_row(y + 1, r1, r2, r0);
        unpack(nxt, r1);
        if (y + 2 >= y1) break;
        nxt = load_raw(y + 4);
        filter_row(y + 2, r2, r0, r1);
        unpack(nxt, r2);
    }
}

} // namespace

namespace {
// Largest input footprint (texels) of any kFsrTW x kFsrTH tile (+halo), with the kernel's own f32 index math.
size_t fsr_max_footprint(const FsrArgs &A, int halo)
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
} // namespace

hipError_t launch_fsr1(const UpscaleLaunch &L, int mode, float easu_sharpness, float rcas_sharpness, bool fast)
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
    // (FAST adds the 3-KiB weight table to the block's LDS: its footprint caps are 192 texels lower; a shape that misses them runs
    // the exact arithmetic, which meets the FAST contract trivially)
    const bool src_lds = mode != 1 && foot <= (mode == 2 ? 1700u : 3072u) - (fast ? 192u : 0u);
    const size_t dyn = src_lds ? foot * sizeof(float4) : 0;
#ifndef NUS_FSR_QUAD
#define NUS_FSR_QUAD 1 // dev macro: 0 = FAST without the quad form at x2 (A/B timing)
#endif
    const bool quad = NUS_FSR_QUAD && fast && L.ow == 2 * L.iw && L.oh == 2 * L.ih; // exactly x2: four output pixels per input texel
    return for_frame_chunks(L, [&](const uint8_t *in, uint8_t *out, uint32_t n) {
        A.in = reinterpret_cast<const uint32_t *>(in);
        A.out = reinterpret_cast<uint32_t *>(out);
        const dim3 block(256), grid(cdiv(L.ow, kFsrTW), cdiv(L.oh, kFsrTH), n);
#define NUS_FSR2(M, V, S) hipLaunchKernelGGL((k_fsr1<M, V, S>), grid, block, dyn, L.stream, A, (int)foot)
#define NUS_FSRF(M, V)                                                                                       \
    do {                                                                                                     \
        if (quad)                                                                                            \
            hipLaunchKernelGGL((k_fsr1<M, V, true, true, true>), grid, block, dyn, L.stream, A, (int)foot);  \
        else                                                                                                 \
            hipLaunchKernelGGL((k_fsr1<M, V, true, true, false>), grid, block, dyn, L.stream, A, (int)foot); \
    } while (0)
#define NUS_FSR(M)                      \
    if (fast && src_lds && vec)         \
        NUS_FSRF(M, true);              \
    else if (fast && src_lds)           \
        NUS_FSRF(M, false);             \
    else if (vec && src_lds)            \
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
#ifndef NUS_RCAS_WALK
#define NUS_RCAS_WALK 1 // dev macro: 0 = RCAS through the LDS tile (rounds 1-4; A/B timing)
#endif
            if (NUS_RCAS_WALK) {
                RcasWalkArgs W;
                W.in = A.in, W.out = A.out, W.w = A.ow, W.h = A.oh, W.px = A.opx, W.sharp = A.rcas_sharp, W.sel = A.sel;
                W.strips = (int)cdiv(L.ow, (uint32_t)kRcasStripCols);
                // row blocks: enough waves to fill the GPU several times over, blocks long enough that the 2 halo rows do not matter
                uint32_t rows = 64;
                while (rows > 16 && (uint64_t)W.strips * cdiv(L.oh, rows) * n < 8192) rows /= 2;
                W.rows_per_block = (int)rows;
                W.row_blocks = (int)cdiv(L.oh, rows);
                const dim3 wblock(256), wgrid(cdiv((uint32_t)(W.strips * W.row_blocks), 4), n);
                const bool wvec = vec && (reinterpret_cast<uintptr_t>(A.in) % 16) == 0 && (reinterpret_cast<uintptr_t>(A.out) % 16) == 0;
                if (wvec)
                    hipLaunchKernelGGL(k_fsr_rcas_walk<true>, wgrid, wblock, 0, L.stream, W);
                else
                    hipLaunchKernelGGL(k_fsr_rcas_walk<false>, wgrid, wblock, 0, L.stream, W);
            } else if (vec)
                NUS_FSR2(FsrMode::Rcas, true, false);
            else
                NUS_FSR2(FsrMode::Rcas, false, false);
        } else {
            NUS_FSR(FsrMode::Fused);
        }
#undef NUS_FSR
#undef NUS_FSRF
#undef NUS_FSR2
    });
}

} // namespace nus

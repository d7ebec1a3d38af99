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

} // namespace nus

// nus_device.hpp -- device helpers shared by the kernel translation units (nus_k_*.hip), plus the
// small host helpers their launchers use.  Everything sits in an anonymous namespace: each unit gets
// its own copy, nothing is exported.
//
// All f32 arithmetic that must match the CPU oracle bit for bit is written with separate multiplies
// and adds; every unit is compiled with -ffp-contract=off and fused multiply-adds appear only where
// spelled __builtin_fmaf.  No MFMA: no stage is a dense contraction.  Pixels are moved as one u32
// each, 16 bytes per lane per access wherever alignment allows.
#pragma once

#include "nus_kernels.hpp"

#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace nus {

namespace {

constexpr int kWave = 64;

// Output pixels are written once and never read back by the kernel that writes them.  Streaming (nt) stores keep them from
// pushing the kernel's INPUT out of the L2 -- a gain for the kernels whose waves re-read input (row-walking resize kernels,
// the replicating nearest / fixed-ratio kernels: -2 ... -26 %), a loss for the packed-u8 bilinear x2 kernel (+5-10 %), the
// blend (+3 %) and the warp (+7 %): chosen per kernel, measured in one run (profiles/r04_nt_stores_by_kernel.txt).
// NUS_NT_STORES = 0 builds plain stores everywhere (A/B knob).
#ifndef NUS_NT_STORES
#define NUS_NT_STORES 1
#endif
typedef uint32_t nus_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t nus_u32x2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void store_out16(uint32_t *dst, const uint4 v)
{
    if constexpr (NT && NUS_NT_STORES)
        __builtin_nontemporal_store(nus_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nus_u32x4 *>(dst));
    else
        *reinterpret_cast<uint4 *>(dst) = v;
}
template <bool NT>
__device__ __forceinline__ void store_out8(uint32_t *dst, const uint2 v)
{
    if constexpr (NT && NUS_NT_STORES)
        __builtin_nontemporal_store(nus_u32x2{v.x, v.y}, reinterpret_cast<nus_u32x2 *>(dst));
    else
        *reinterpret_cast<uint2 *>(dst) = v;
}

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

// p[i] for a wave-uniform index, as a SCALAR load.  Written as `readfirstlane(p[i])` the compiler issues a vector load and
// waits for vmcnt(0) before the value can be used -- and vector memory operations retire in order, so every store the wave has
// in flight is drained with it, once per output row in the row-walking kernels (round 3: tools/check_hidden_loads.py lists such
// waits; k_lanczos3_r32 / _xs / k_resize_win / k_bilinear_table / k_nearest_table all had one).
template <typename T>
__device__ __forceinline__ T uniform_load(const T *p, size_t i)
{
    typedef const __attribute__((address_space(4))) T *cptr;
    return ((cptr)(uintptr_t)p)[i];
}

template <bool EXACT>
__device__ __forceinline__ float mac(float acc, float v, float w)
{
    if (EXACT) return acc + v * w;   // two roundings, as the CPU restatement
    return __builtin_fmaf(v, w, acc); // one rounding
}

// The same, with the EXACT product consumed where it is formed: left to itself the compiler batches the products of a tap
// loop ahead of the additions (k_lanczos3_x2 EXACT: 310 VGPRs, one wave per SIMD; with this 234, two waves, 29.9 -> 23.0 us per
// 1080p -> 4K frame; x3/2 36.6 -> 27.5).  Not for the x3 / x4 kernel, which is faster with the batches (profiles/
// r02_lanczos_exact_mode_products.txt).
template <bool EXACT, bool PIN_FMA = false>
__device__ __forceinline__ float mac_tight(float acc, float v, float w)
{
    if (EXACT) {
        float p = v * w;
        asm volatile("" : "+v"(p));
        return acc + p;
    }
    // PIN_FMA: the fused operations in program order too (one tap chain after the other).  k_lanczos3_r43: 192 -> 153 VGPRs,
    // a third wave per SIMD, 6.7 -> 6.1 us per 1080p -> 1440p frame; the x3/2 kernel needs MORE registers that way, x2 the same.
    float r = __builtin_fmaf(v, w, acc);
    if (PIN_FMA) asm volatile("" : "+v"(r));
    return r;
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



__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Correctly rounded x / y from z = RN(1/y) with one multiply and two FMAs (Markstein): q = RN(x z),
// r = x - q y (exact in the FMA), result = RN(q + r z).  Equal to the IEEE quotient for every y whose
// mantissa is not all ones (Markstein's theorem) and, checked exhaustively over every mantissa of x on the
// CPU (tests/test_div_by_recip.py), also for the all-ones mantissa -- the theorem's exception is not needed
// for this sequence -- and for y = 9 and y = 255, as long as quotient and remainder stay in the normal range
// (the sequence is invariant under scaling x or y by powers of two).  Replaces ~11 slow-class instructions
// per division by 3 fast ones.
__device__ __forceinline__ float div_by_recip(float x, float y, float z)
{
    const float q = x * z;
    const float r = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(r, z, q);
}

// u8 -> f32 / 255 (the Rgba8Unorm view of a frame); exact IEEE quotient via the reciprocal + 2 FMAs
// (equal to x / 255.0f for all 256 inputs, checked on the CPU).
__device__ __forceinline__ float4 unorm8(uint32_t p)
{
    const float z = 1.0f / 255.0f;
    return make_float4(div_by_recip(ch_f32(p, 0), 255.0f, z), div_by_recip(ch_f32(p, 1), 255.0f, z),
                       div_by_recip(ch_f32(p, 2), 255.0f, z), div_by_recip(ch_f32(p, 3), 255.0f, z));
}

// trunc((1-t) a + t b) per channel: the zero-flow in-between pixel (interpolation/mod.rs:407-411)
__device__ __forceinline__ uint32_t blend_px(uint32_t a, uint32_t b, float t, float nt)
{
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) o = pack_trunc_u8(nt * ch_f32(a, c) + t * ch_f32(b, c), c, o);
    return o;
}

// The value a neighbouring lane holds, by a DPP move (no LDS round trip): wave_up = lane - 1's, wave_down = lane + 1's.
// Lane 0 / lane 63, which have no such neighbour, get 0 -- the row-walking kernels keep those lanes as halo.
__device__ __forceinline__ float wave_up(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_down(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, true));
}

// XCD-aware workgroup order.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs (blocks b and
// b + 8 share one, observed behaviour, speed only), each with its own L2.  This bijective remap hands every XCD a
// CONTIGUOUS range of virtual workgroup ids, so that workgroups which re-read each other's halo rows / columns
// run behind the same L2.  Returns the virtual linear id of this workgroup in a grid of `nwg` workgroups.
__device__ __forceinline__ uint32_t xcd_contiguous_id(uint32_t orig, uint32_t nwg)
{
    const uint32_t q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
}

// The same for a 3-D grid (x fastest, as the dispatcher walks it): the (x, y, z) this workgroup should work on.
struct GridPos {
    uint32_t x, y, z;
};
__device__ __forceinline__ GridPos xcd_contiguous_pos()
{
    const uint32_t lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const uint32_t vid = xcd_contiguous_id(lin, gridDim.x * gridDim.y * gridDim.z);
    const uint32_t xy = gridDim.x * gridDim.y, r = vid % xy;
    return GridPos{r % gridDim.x, r / gridDim.x, vid / xy};
}

constexpr uint32_t kMaxGridZ = 65535;

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Frames go on grid.z (<= 65535 per launch); longer batches are issued in chunks: f(in, out, n) per chunk.
inline size_t launch_in_frame_bytes(const UpscaleLaunch &L) { return L.in_stride ? L.in_stride : (size_t)L.iw * L.ih * 4; }
// index of the chunk's first frame (for kernels that take a second input per frame), from the chunk's input pointer
inline size_t chunk_first_frame(const UpscaleLaunch &L, const uint8_t *chunk_in)
{
    return (size_t)(chunk_in - L.in) / launch_in_frame_bytes(L);
}
template <typename F>
hipError_t for_frame_chunks(const UpscaleLaunch &L, F &&f)
{
    const size_t in_bytes = launch_in_frame_bytes(L), out_bytes = (size_t)L.ow * L.oh * 4;
    for (uint32_t done = 0; done < L.n_frames;) {
        const uint32_t n = L.n_frames - done < kMaxGridZ ? L.n_frames - done : kMaxGridZ;
        f(L.in + (size_t)done * in_bytes, L.out + (size_t)done * out_bytes, n);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += n;
    }
    return hipSuccess;
}

} // namespace

} // namespace nus

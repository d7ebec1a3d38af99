// nus_kernels.hpp -- launch interface between the host classes and the gfx950 kernels.
// Everything here is device-pointer based; no allocation, no synchronisation
// (safe to capture into a hipGraph).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nus {

// Device-side tables of one initialised upscaler (built on the host by nus_tables.cpp).
struct DeviceTables {
    // nearest: source index per output index
    const uint32_t *nn_sx = nullptr, *nn_sy = nullptr;
    // bilinear: i0 and fraction per output index
    const uint32_t *bl_x0 = nullptr, *bl_y0 = nullptr;
    const float *bl_fx = nullptr, *bl_fy = nullptr;
    // lanczos general: tap window per output index, weights [n][taps_stride]
    const int32_t *lz_lx = nullptr, *lz_ly = nullptr;
    const uint32_t *lz_nx = nullptr, *lz_ny = nullptr;
    const float *lz_wx = nullptr, *lz_wy = nullptr;
    uint32_t lz_stride = 0;
    // down-scaling stream kernel: per input row 7 slot weights + completion word, and the completing row per output row
    const uint32_t *lz_down_rows = nullptr;
    const int32_t *lz_down_done = nullptr;
    // lanczos x2 fast path: per-output-row weights in the 6-tap phase frame [oh][6]
    const float *lz_wy6 = nullptr;
    const float *lz_wx6 = nullptr; // P/Q kernel: per-output-column weights in the 6-tap phase frame [ow][6]
    float lz_wxe[6] = {0}, lz_wxo[6] = {0}; // interior horizontal weights, even / odd outputs
    float lz_wx_left[48] = {0}, lz_wx_right[48] = {0}; // phase-frame weights of the 8 edge outputs per side
    float lz_wxs[4][6] = {{0}};                        // integer factors x3 / x4: interior weights per phase
    float lz_wxs_left[16][6] = {{0}}, lz_wxs_right[16][6] = {{0}}; // ... and of the 4 S edge outputs per side
    // x3: interior weights by class of the input index (nus_tables.hpp: lanczos_xs_weight_classes); null at x4
    const uint32_t *lz_xs_cls_x = nullptr, *lz_xs_cls_y = nullptr; // [iw], [ih]
    const float *lz_xs_wcls_x = nullptr, *lz_xs_wcls_y = nullptr;  // [classes][S][6]
};

// v_perm_b32 selectors applied to every loaded input pixel: RGBA8 as is, or BGRA8 (capture order,
// nu_scaler_core/src/lib.rs:251-270) swizzled to RGBA on the way in.
constexpr uint32_t kSelRGBA = 0x03020100u, kSelBGRA = 0x03000102u;
// ... and the X variants (alpha byte undefined, as BGRX capture surfaces): selector byte 0x0D yields 0xFF, so the
// pixel enters every kernel opaque and the output alpha is 255.
constexpr uint32_t kSelRGBX = 0x0D020100u, kSelBGRX = 0x0D000102u;
inline uint32_t input_selector(int format) // nus_pixel_format
{
    switch (format) {
    case 1: return kSelBGRA;
    case 2: return kSelRGBX;
    case 3: return kSelBGRX;
    default: return kSelRGBA;
    }
}

struct UpscaleLaunch {
    const uint8_t *in = nullptr; // n_frames contiguous frames
    uint8_t *out = nullptr;
    uint32_t iw = 0, ih = 0, ow = 0, oh = 0;
    uint32_t n_frames = 1;
    hipStream_t stream = nullptr;
    // x2 resize kernels only: input frame stride in bytes (0 = tightly packed), and an optional
    // second frame per unit to blend with on the fly (zero-flow in-between frame at blend_t)
    size_t in_stride = 0;
    const uint8_t *in_b = nullptr;
    size_t in_b_stride = 0;
    float blend_t = 0.5f;
    uint32_t in_sel = kSelRGBA; // input channel order
    // Exact-x2 kernels only (nearest, bilinear, the resize filters): input rows [row0, row0 + rows) of the frame instead of all
    // of it (rows == 0: all).  The kernels still see the whole frame -- tap rows beyond the range are read, not clamped -- so the
    // host path can upscale a frame band by band while the rest of it is still on the bus (HipUpscaler::upscale).  row0 must be
    // a multiple of the kernel's rows per wave (launch_lanczos_x2's rows_per_wave; 4 for the others).
    uint32_t row0 = 0, rows = 0;
};

// Kernel variants (chosen once at initialize).
enum class Variant : int {
    NearestTable = 0, // any scale, index tables
    NearestX2,        // exact x2, 16-B loads/stores
    NearestRatio,     // exact x3/2, x4/3, x3, x4: an input group per lane copied into P outputs, a row group into P rows
    BilinearTable,    // any scale, f32, CPU or WGSL arithmetic
    BilinearX2Int,    // exact x2, CPU arithmetic done in packed-u8 integer ops
    BilinearRatio,    // exact x3/2, x4/3, x3, x4, CPU form: one input group per lane, P outputs, row groups
    LanczosGeneral,   // any scale, direct separable evaluation per output pixel (fallback)
    ResizeRows,       // any scale, separable: V pass into an LDS row, H pass out of it
    ResizeWin,        // up-scaling: V pass from a register row window (as the x2 kernel), H pass through the LDS row
    ResizeDown,       // down-scaling: input rows streamed once into the vertical sums of the 7 output rows in flight
    LanczosX2RegWin,  // exact x2, register sliding window + wave shifts
    LanczosXsRegWin,  // exact x3 / x4, same design with S output rows per input row
    LanczosR32RegWin, // exact x3/2, same design: three output rows per pair of input rows
    LanczosR43RegWin, // exact x4/3, same design: four output rows per group of three input rows
    LanczosPqRegWin,  // x5/4, x6/5, x5/3, x5/2, same design: P output rows per group of Q input rows, weights from the tables
    FsrEasu,          // FSR1-style EASU alone (any scale)
    FsrRcas,          // FSR1-style RCAS alone (same size in and out)
    Fsr1Fused,        // EASU tile (+1 px halo) in LDS, RCAS out of it
    Fsr1TwoPass,      // EASU into a scratch image in HBM, the row-walking RCAS out of it (frames of >= 1 MiB: faster than the fused tile)
};

const char *variant_name(Variant v);

hipError_t launch_nearest_table(const UpscaleLaunch &L, const DeviceTables &T);
hipError_t launch_nearest_x2(const UpscaleLaunch &L);
hipError_t launch_nearest_ratio(const UpscaleLaunch &L); // exact x3/2, x4/3, x3, x4 (host-checked table shape: source index Q (o / P) + (o % P) Q / P)
hipError_t launch_bilinear_table(const UpscaleLaunch &L, const DeviceTables &T, bool wgsl_form);
hipError_t launch_bilinear_x2_int(const UpscaleLaunch &L);
// exact x3/2, x4/3, x3, x4, CPU form (host-checked table shape: i0 = Q (o / P) + (o % P) Q / P, fraction 0 where (o % P) Q % P == 0)
hipError_t launch_bilinear_ratio(const UpscaleLaunch &L, const DeviceTables &T);
// edge_only: evaluate only the first and last `edge_cols` output columns.
hipError_t launch_lanczos_general(const UpscaleLaunch &L, const DeviceTables &T, bool exact,
                                  uint32_t edge_cols);
// ncols_max: widest input-column footprint of a 64*N-column output segment (LDS row length);
// small_taps: every tap window has <= 8 taps.
// union_taps: widest union of the tap windows of 4 adjacent outputs (x % 4 == 0), 0 = unknown / do not use.
hipError_t launch_resize_rows(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max,
                              bool small_taps, uint32_t union_taps);
// Up-scaling variant (ow % 4 == 0, <= 7 vertical / <= 8 horizontal taps, first tap row advancing by <= 1
// per output row, ncols_max <= 192): vertical taps from a register window.
// outputs_per_lane: 4 (segments of 256 output columns) or 2 (segments of 128); ncols_max and union_taps are for that width.
hipError_t launch_resize_win(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max,
                             uint32_t union_taps, uint32_t outputs_per_lane);
// Down-scaling variant (nus_k_resize_down.hip): needs T.lz_down_rows / lz_down_done (build_down_stream_tables
// succeeded: 7 accumulator slots suffice), seg_w = output columns per wave (<= 64), ncols_max = widest footprint of a
// seg_w-column output segment <= 320.  max_taps_x: widest horizontal window (<= 32).
hipError_t launch_resize_down(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t ncols_max,
                              uint32_t max_taps_x, uint32_t seg_w);
// main x2 kernel only: the first / last kLanczosX2EdgeCols output columns are NOT written;
// follow it with launch_lanczos_x2_edges(L, T, exact).
hipError_t launch_lanczos_x2(const UpscaleLaunch &L, const DeviceTables &T, bool exact,
                             uint32_t rows_per_wave);

hipError_t launch_lanczos_x2_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact);
// One pipeline step over n_frames pairs (L.in = A_k, L.in_b = B_k) in ONE launch of the x2 kernel: L.out = the up-scaled real
// frames, out_mid = the up-scaled in-between frames (blend of the pair at L.blend_t), mid = the in-between frames themselves
// (n_frames tightly packed input-size frames; may be null).  order: 0 = frame-major, 1 = row-block-major wave order.
// Main kernel only: follow it with launch_lanczos_x2_edges for both outputs.
struct UnitOutputs {
    uint8_t *out_mid = nullptr;
    uint8_t *mid = nullptr;
    uint32_t order = 1;
};
hipError_t launch_lanczos_x2_unit(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave,
                                  const UnitOutputs &U);
// exact x3 / x4 (factor): main kernel only, the first / last 4 * factor output columns are NOT written;
// follow it with launch_lanczos_xs_edges(L, T, exact, factor).
hipError_t launch_lanczos_xs(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t factor,
                             uint32_t rows_per_wave);
hipError_t launch_lanczos_xs_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t factor);
// exact x3/2 (2 ow == 3 iw, 2 oh == 3 ih, iw % 8 == 0, ih even): main kernel only, the first / last 12 output columns are
// NOT written; follow it with launch_lanczos_r32_edges(L, T, exact).
hipError_t launch_lanczos_r32(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave);
hipError_t launch_lanczos_r32_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact);
// exact x4/3 (3 ow == 4 iw, 3 oh == 4 ih, iw % 12 == 0): main kernel only, the first / last 8 output columns are NOT written;
// follow it with launch_lanczos_r43_edges(L, T, exact).
hipError_t launch_lanczos_r43(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave);
hipError_t launch_lanczos_r43_edges(const UpscaleLaunch &L, const DeviceTables &T, bool exact);
// x5/4, x6/5, x7/5, x8/5, x5/3, x5/2, x7/2 (Q ow == P iw, Q oh == P ih, iw % Q == 0, ih % Q == 0, ow % 4 == 0; T.lz_wx6 / T.lz_wy6: the tables in frame form,
// nus_tables.hpp: lanczos_pq_phase_frame).  Writes every output column: no edge pass.
bool lanczos_pq_supported(uint32_t P, uint32_t Q);
uint32_t lanczos_pq_strip_cols(uint32_t P, uint32_t Q); // input columns per wave
hipError_t launch_lanczos_pq(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t P, uint32_t Q, uint32_t rows_per_wave,
                             bool narrow);
// FSR1-style passes (fsr.rs:24-260).  mode 0: EASU, 1: RCAS (iw == ow, ih == oh), 2: EASU then RCAS fused.
// fast: EASU in FAST arithmetic where the LDS source tile applies (nus_k_fsr.hip; option "fsr_fast")
hipError_t launch_fsr1(const UpscaleLaunch &L, int mode, float easu_sharpness, float rcas_sharpness, bool fast = false);

constexpr uint32_t kLanczosX2EdgeCols = 8; // output columns left to the general kernel per side
#ifndef NUS_LZ_STRIP_COLS
#define NUS_LZ_STRIP_COLS 240 // 60 storing lanes: a strip's output rows are whole 128-B lines (x2: 1920 B = 15 lines; 1080p is 8
                             // strips exactly); 248 = all 62 non-halo lanes storing, strip joints in mid-line (dev macro, A/B)
#endif
constexpr uint32_t kLanczosX2StripCols = NUS_LZ_STRIP_COLS; // input columns produced per wave (lanes 1 .. StripCols / 4, 4 columns each)

struct WarpLaunch {
    const uint8_t *a = nullptr, *b = nullptr;
    const float *flow = nullptr; // nullptr: zero flow
    bool flow_half = false;      // the flow field is 2 x f16 per pixel (Rg16Float, wgpu_interpolator.rs:276), not 2 x f32
    bool fma = false;            // dense-flow warp in fused multiply-adds (+-1 LSB) instead of the CPU's separate roundings
    uint8_t *out = nullptr;
    size_t a_stride = 0, b_stride = 0; // bytes between consecutive pairs
    uint32_t w = 0, h = 0;
    float t = 0.5f;
    uint32_t n_pairs = 1;
    hipStream_t stream = nullptr;
    uint32_t in_sel = kSelRGBA; // channel order of both input frames (the output is RGBA)
};

hipError_t launch_warp_blend(const WarpLaunch &L);

// BGRA -> RGBA (in place allowed: in == out).
hipError_t launch_swizzle_bgra(const uint8_t *in, uint8_t *out, size_t npx, hipStream_t stream);

// Optical-flow front end (f32 RGBA images, float2 flows; device pointers).
hipError_t launch_rgba8_to_f32(const uint8_t *in, float *out, uint32_t w, uint32_t h, hipStream_t stream);
hipError_t launch_blur(const float *in, float *out, uint32_t w, uint32_t h, bool horizontal, hipStream_t stream);
hipError_t launch_downsample(const float *in, float *out, uint32_t w, uint32_t h, hipStream_t stream);
hipError_t launch_horn_schunck(const float *i1, const float *i2, const float *flow_in, float *flow_out, uint32_t w,
                               uint32_t h, float lambda, hipStream_t stream);
// The launchers below take a batch: `n` independent images / pairs on the grid's z axis, buffer b of item z at
// b + z * stride (strides in elements of the buffer's type; bytes for an RGBA8 input).  n = 1 ignores the strides.
// blur H + blur V + downsample of one level in one launch (LDS tile with a 2-pixel halo); the level
// itself is written as its luminance plane (w*h floats), which is all Horn-Schunck reads of it.
// `kernel`: LDS-tile or register-pipelined ("streamed") form of the pyramid and the multi-step Jacobi kernels
enum JacobiKernel { kJacobiAuto = 0, kJacobiTiles = 1, kJacobiStream = 2, kJacobiStreamFast = 3 }; // Fast: k_hs_stream_fast, every level
hipError_t launch_pyramid_level(const void *in, bool u8_input, float *level_lum, float *next, uint32_t w, uint32_t h,
                                hipStream_t stream, uint32_t n = 1, size_t in_stride = 0, size_t lum_stride = 0,
                                size_t next_stride = 0, int kernel = 0);
// Fast path of the same iteration: derivatives once per level, then K steps per launch in LDS.
// i1 / i2: f32 RGBA level images, or their luminance planes (luminance_planes).
hipError_t launch_pyramid_level_fast(const void *in, bool u8_input, float *level_lum, float *next, uint32_t w, uint32_t h,
                                     hipStream_t stream, uint32_t n, size_t in_stride, size_t lum_stride, size_t next_stride);
hipError_t launch_hs_prepare(const float *i1, const float *i2, bool luminance_planes, float *coef, uint32_t w, uint32_t h,
                             hipStream_t stream, uint32_t n = 1, size_t img_stride = 0, size_t coef_stride = 0);
// launch_hs_prepare on luminance planes + launch_flow_upsample of the coarser level's flow, one launch.
hipError_t launch_hs_level_setup(const float *l1, const float *l2, float *coef, uint32_t w, uint32_t h, const float *coarse,
                                 uint32_t cw, uint32_t ch, float *flow, float scale, hipStream_t stream, uint32_t n = 1,
                                 size_t lum_stride = 0, size_t coef_stride = 0, size_t coarse_stride = 0,
                                 size_t flow_stride = 0);
// HsWarp (round 5; used only with NUS_HS_FUSED_WARP=1 in the environment: measured slower than the warp kernel behind the estimator):
// the LAST launch of the finest level can warp + blend the pair's two frames with the flow it has just finished
// (dense-flow warp in FMA mode, nus_warp_device.hpp) and store the in-between frame -- the flow then never has to be written for the
// warp to read it back.  frames: RGBA8 frame of pair z at frames + z * frame_stride bytes, its partner one frame_stride further;
// mid: w * h RGBA8 pixels per pair, tightly packed.  *warped tells the caller whether the launch that ran could do it (FAST ring
// form, not the launch that upsamples the coarser level, frames of at least 2 x 2 pixels and less than 4 GiB); if not, the caller
// runs the warp kernel itself.  With a warp and final_out == nullptr the flow of the last launch is not stored at all.
struct HsWarp {
    const uint8_t *frames = nullptr;
    size_t frame_stride = 0;
    uint8_t *mid = nullptr;
    float t = 0.5f;
    uint32_t sel = 0;
    // the level's FINAL flow as 2 x IEEE half per cell (round to nearest even) instead of 2 x f32: the reference's live flow layout,
    // Rg16Float (wgpu_interpolator.rs:276), for a warp that reads it that way.  Honoured by the FAST streamed kernel's last launch
    // (*wrote_half says so); everything else leaves f32 and the caller converts (launch_flow_to_half).
    uint32_t out_half = 0;
    // dev switch NUS_HS_L0_HALF_BETWEEN=1 (round 6, measured: profiles/r06_flow_level0_half_between_launches.txt): the flow BETWEEN the
    // finest level's two launches as Rg16Float as well -- the launch reads 2 x half per cell
    uint32_t in_half = 0;
};
// n cells of 2 x f32 -> 2 x f16 (round to nearest even)
hipError_t launch_flow_to_half(const float *src, void *dst, size_t n_cells, hipStream_t stream);
hipError_t launch_hs_iterate(const float *coef, float lambda, float **flow_a, float **flow_b, uint32_t w, uint32_t h,
                             uint32_t iterations, bool zero_start, float *final_out, hipStream_t stream, uint32_t n = 1,
                             size_t coef_stride = 0, size_t flow_stride = 0, size_t final_stride = 0, int kernel = 0,
                             const float *lum1 = nullptr, size_t lum_stride = 0, const float *coarse = nullptr, uint32_t cw = 0,
                             uint32_t ch = 0, float coarse_scale = 0.0f, size_t coarse_stride = 0, const HsWarp *warp = nullptr,
                             bool *warped = nullptr, bool *wrote_half = nullptr);
bool hs_iterate_streams(uint32_t w, uint32_t h, uint32_t n, int kernel);
hipError_t launch_flow_upsample(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh,
                                float scale, hipStream_t stream, uint32_t n = 1, size_t src_stride = 0, size_t dst_stride = 0);

// Box calibration (nus_k_probe.hip; bench.py's denominators, not on the product path): kind 0 hipMemcpyDtoDAsync, 1 stream
// copy, 2 write-only, 3 read-only (d_dst: 4 bytes of device memory), 4 one read : four writes (d_dst holds 4 * bytes),
// 5 VGPR-only FMA chains (kProbeValuBlocks blocks of 256 lanes, 16 chains of `iters` FMAs each; d_dst: that many floats).
constexpr uint32_t kProbeValuBlocks = 2048;
hipError_t launch_probe(int kind, const void *d_src, void *d_dst, size_t bytes, uint32_t iters, hipStream_t stream);

} // namespace nus

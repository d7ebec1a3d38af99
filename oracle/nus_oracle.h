/*
 * nus_oracle.h -- CPU oracle for the NU_Scaler upscale + interpolation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nu_scaler_amd/ (the product) may
 * include, link, import or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * Every function is a restatement (not a copy) of the reference arithmetic it
 * cites; see nus_oracle.c for the per-function citations and pinning status.
 */
#ifndef NUS_ORACLE_H
#define NUS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* All images: tightly packed RGBA8, row-major, 4 bytes per pixel. */

/* Nearest neighbour.  Nu_scale/src/upscale/common.rs:188-198. */
void orc_nearest(const uint8_t *in, uint32_t iw, uint32_t ih,
                 uint8_t *out, uint32_t ow, uint32_t oh);

/* Bilinear, CPU form (THE bilinear oracle).  Nu_scale/src/upscale/common.rs:199-231. */
void orc_bilinear(const uint8_t *in, uint32_t iw, uint32_t ih,
                  uint8_t *out, uint32_t ow, uint32_t oh);

/* Bilinear, WGSL form (diff-only variant).  nu_scaler_core/src/upscale/mod.rs:209-263. */
void orc_bilinear_wgsl(const uint8_t *in, uint32_t iw, uint32_t ih,
                       uint8_t *out, uint32_t ow, uint32_t oh);

/* Separable resize as done by image-0.24.9 imageops::resize (third-party
 * dependency, not vendored in the reference; PARITY UNPINNED).
 * filter: 0 = Lanczos3, 1 = CatmullRom, 2 = Triangle.  Returns 0 or -1. */
int orc_resize(const uint8_t *in, uint32_t iw, uint32_t ih,
               uint8_t *out, uint32_t ow, uint32_t oh, int filter);
int orc_lanczos3(const uint8_t *in, uint32_t iw, uint32_t ih,
                 uint8_t *out, uint32_t ow, uint32_t oh);

/* One axis of the resize weight computation (exposed so the product's host
 * tables can be diffed against it).  weights is out_n x max_taps, left and
 * ntaps are out_n.  Returns the largest ntaps, or -1 if it exceeds max_taps. */
int orc_resize_axis(uint32_t in_n, uint32_t out_n, int filter, uint32_t max_taps,
                    int32_t *left, uint32_t *ntaps, float *weights);

/* Warp + blend.  Geometry: shaders/warp_blend.wgsl:25-43; sampling + rounding:
 * nu_scaler_core/src/interpolation/mod.rs:467-510 and :386-411.
 * flow == NULL means zero flow (the live reference behaviour). flow is 2 floats
 * per pixel (dx, dy), pixel delta from frame A to frame B. */
void orc_warp_blend(const uint8_t *a, const uint8_t *b, const float *flow,
                    uint32_t w, uint32_t h, float t, uint8_t *out);

/* ---- optical-flow front end ("next" row) --------------
 * Images are f32 RGBA (4 floats per pixel), flows 2 floats per pixel (dx, dy).
 * u8 -> f32: value / 255.0f (the Rgba8Unorm view the interpolator has of its frames). */
void orc_rgba8_to_f32(const uint8_t *in, uint32_t w, uint32_t h, float *out);
/* shaders/gaussian_blur_h.wgsl:22-52 / gaussian_blur_v.wgsl:22-52: 5 taps [1,4,6,4,1]/16, clamped. */
void orc_blur_h(const float *in, uint32_t w, uint32_t h, float *out);
void orc_blur_v(const float *in, uint32_t w, uint32_t h, float *out);
/* shaders/downsample.wgsl:14-38: 2x2 box average into ((w+1)/2, (h+1)/2)
 * (wgpu_interpolator.rs:1008-1009); source reads beyond the edge clamp (build-defined:
 * WGSL leaves out-of-bounds textureLoad open). */
void orc_downsample(const float *in, uint32_t w, uint32_t h, float *out);
/* shaders/horn_schunck.wgsl:48-92: one Jacobi step. */
void orc_horn_schunck_step(const float *i1, const float *i2, const float *flow_in,
                           uint32_t w, uint32_t h, float lambda, float *flow_out);
/* shaders/flow_upsample.wgsl:21-37: bilinear, normalised-UV sampling (half-pixel centres,
 * clamp to edge); vectors multiplied by `scale` (1.0 = the shader as written). */
void orc_flow_upsample(const float *src, uint32_t sw, uint32_t sh,
                       float *dst, uint32_t dw, uint32_t dh, float scale);
/* Composite (host logic of wgpu_interpolator.rs:969-1203 + build-defined coarse-to-fine
 * warm start): pyramids of both frames, `coarse_iters` steps from zero flow at the coarsest
 * level, then per finer level: upsample x2 (vectors x2) and `refine_iters` more steps.
 * flow_out: w*h*2 floats at full resolution.  Returns 0 or -1. */
int orc_flow_estimate(const uint8_t *a, const uint8_t *b, uint32_t w, uint32_t h,
                      uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters,
                      float lambda, float *flow_out);

/* ---- FSR1-style EASU + RCAS ("next" row; PARITY UNPINNED) ----
 * nu_scaler_core/src/upscale/fsr.rs:24-169 (EASU: 4x4 taps, cubic weight of the distance
 * projected on the local gradient direction, optional mix toward the centre texel) and
 * :173-260 (RCAS: 5-tap laplacian sharpen, strength faded by local luma contrast).
 * Both write alpha = 255 and pack by truncation. */
void orc_fsr_easu(const uint8_t *in, uint32_t iw, uint32_t ih,
                  uint8_t *out, uint32_t ow, uint32_t oh, float sharpness);
void orc_fsr_rcas(const uint8_t *in, uint32_t w, uint32_t h, uint8_t *out, float sharpness);
/* EASU into a temporary, then RCAS.  Returns 0 or -1 (allocation). */
int orc_fsr1(const uint8_t *in, uint32_t iw, uint32_t ih, uint8_t *out, uint32_t ow, uint32_t oh,
             float easu_sharpness, float rcas_sharpness);

/* OpenMP row-parallel variants for the "all host cores" baseline
 * (same arithmetic, rows distributed over threads).  threads<=0: all cores. */
void orc_nearest_mt(const uint8_t *in, uint32_t iw, uint32_t ih,
                    uint8_t *out, uint32_t ow, uint32_t oh, int threads);
void orc_bilinear_mt(const uint8_t *in, uint32_t iw, uint32_t ih,
                     uint8_t *out, uint32_t ow, uint32_t oh, int threads);
int orc_lanczos3_mt(const uint8_t *in, uint32_t iw, uint32_t ih,
                    uint8_t *out, uint32_t ow, uint32_t oh, int threads);
/* any filter of orc_resize; row blocks per thread, bit-identical to orc_resize (tests/test_oracle_golden.py) */
int orc_resize_mt(const uint8_t *in, uint32_t iw, uint32_t ih,
                  uint8_t *out, uint32_t ow, uint32_t oh, int filter, int threads);
void orc_warp_blend_mt(const uint8_t *a, const uint8_t *b, const float *flow,
                       uint32_t w, uint32_t h, float t, uint8_t *out, int threads);
int orc_max_threads(void);

/* Synthetic inputs. */
/* S1: nu_scaler_core/src/benchmark.rs:188-207 gradient, shifted by `shift` columns
 * (frame k of the synthetic stream uses shift = k). */
void orc_gen_gradient(uint8_t *out, uint32_t w, uint32_t h, uint32_t shift);
/* S3: uniform u8 noise from splitmix64(seed). */
void orc_gen_noise(uint8_t *out, uint32_t w, uint32_t h, uint64_t seed);
/* S4: nu_scaler_py/test_interpolator.py:23-34 centred box, given RGBA colour. */
void orc_gen_box(uint8_t *out, uint32_t w, uint32_t h,
                 uint8_t r, uint8_t g, uint8_t b, uint8_t a);

#ifdef __cplusplus
}
#endif
#endif

/*
 * nuscaler_hip.h -- C ABI of the MI355X (gfx950) upscale + frame-interpolation path.
 *
 * This is the drop-in boundary for nu_scaler_core's per-pixel hot path: a Rust shim
 * (`impl Upscaler for HipUpscaler`) or the ctypes module
 * nu_scaler_amd binds exactly these entry points.  Plain pointers and sizes only.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   trait Upscaler                      nu_scaler_core/src/upscale/mod.rs:67-88
 *   UpscalerFactory::create_upscaler    nu_scaler_core/src/upscale/mod.rs:95-117
 *   WgpuUpscaler::upscale / _batch      nu_scaler_core/src/upscale/mod.rs:935-1058, :609-640
 *   WgpuFrameInterpolator::interpolate_py   nu_scaler_core/src/wgpu_interpolator.rs:215-491
 *   get_last_gpu_duration_ms            nu_scaler_core/src/wgpu_interpolator.rs:494-497
 *   trait FrameInterpolator (shape)     nu_scaler_core/src/interpolation/mod.rs:29-44
 *
 * Frames are tightly packed RGBA8, row-major.  Every function returns NUS_OK (0)
 * or a negative nus_status; the message is available from nus_*_last_error(handle)
 * (per handle) and nus_last_error() (thread-local, also covers create failures).
 * Concurrent calls on one handle are serialised by a per-handle mutex (the reference
 * calls upscale(&self) from rayon threads, upscale/mod.rs:619-624).
 * There is no CPU fallback: without a usable HIP device the compute entry points
 * fail with NUS_ERR_NO_DEVICE.
 * Host entry points that take pageable buffers copy through pinned staging memory; copies of
 * 1 MiB and more are split over a few process-wide helper threads started on first use (up to 6, fewer where the process's
 * CPU mask or cgroup quota is smaller; environment: NUS_COPY_THREADS=n, 0 = copy on the calling thread only).  The same
 * threads make the pages of a pageable OUTPUT buffer present while its frame is on the GPU (a result buffer fresh from the
 * allocator -- the Vec / PyBytes of the trait's `upscale` -- otherwise takes its first-touch faults inside the copy-out);
 * buffer contents are never touched before the frame's bytes arrive.  A fresh buffer that is its own mapping also gets a transparent-huge-page
 * hint (environment NUS_NO_THP_HINT=1 turns that off).
 */
#ifndef NUSCALER_HIP_H
#define NUSCALER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NUS_ABI_VERSION 1

typedef enum nus_status {
    NUS_OK = 0,
    NUS_ERR_INVALID_ARGUMENT = -1,
    NUS_ERR_NOT_INITIALIZED = -2, /* "Upscaler not initialized. Call initialize() first." (mod.rs:937-939) */
    NUS_ERR_SIZE_MISMATCH = -3,   /* "Input data size (..) does not match expected input buffer size (..)" (mod.rs:960-966) */
    NUS_ERR_HIP = -4,
    NUS_ERR_NO_DEVICE = -5,
    NUS_ERR_UNSUPPORTED = -6,
    NUS_ERR_OUT_OF_MEMORY = -7
} nus_status;

/* UpscaleAlgorithm (mod.rs:50-53) plus the build-defined Lanczos-3 value. */
typedef enum nus_algorithm {
    NUS_ALG_NEAREST = 0,
    NUS_ALG_BILINEAR = 1,
    NUS_ALG_LANCZOS3 = 2,
    /* "next" row: the other image-0.24.9 filters the legacy
     * BasicUpscaler delegates to (Nu_scale/src/upscale/common.rs:233-260) */
    NUS_ALG_BICUBIC = 3,  /* FilterType::CatmullRom (UpscalingAlgorithm::Bicubic) */
    NUS_ALG_TRIANGLE = 4, /* FilterType::Triangle (what Lanczos2 / Mitchell map to there) */
    /* same row: the FSR1-style shader pair the reference keeps but never dispatches
     * (nu_scaler_core/src/upscale/fsr.rs:24-169 EASU, :173-260 RCAS).  PARITY UNPINNED. */
    NUS_ALG_FSR1 = 5,     /* EASU then RCAS, fused (the EASU image stays in LDS) */
    NUS_ALG_FSR_EASU = 6, /* EASU alone */
    NUS_ALG_FSR_RCAS = 7  /* RCAS alone: output size must equal input size */
} nus_algorithm;

/* UpscalingQuality (mod.rs:37-46), same order.  Quality never changes the arithmetic
 * (mod.rs:1072-1077). */
typedef enum nus_quality {
    NUS_QUALITY_ULTRA_PERFORMANCE = 0,
    NUS_QUALITY_ULTRA = 1,
    NUS_QUALITY_QUALITY = 2,
    NUS_QUALITY_BALANCED = 3,
    NUS_QUALITY_PERFORMANCE = 4,
    NUS_QUALITY_NATIVE = 5
} nus_quality;

/* UpscalingTechnology (mod.rs:56-64), same order. */
typedef enum nus_technology {
    NUS_TECH_NONE = 0,
    NUS_TECH_FSR = 1,
    NUS_TECH_DLSS = 2,
    NUS_TECH_WGPU = 3,
    NUS_TECH_FALLBACK = 4
} nus_technology;

/* WorkgroupSizePreset (wgpu_interpolator.rs:98-127).  Accepted for API parity; the
 * HIP kernels pick their own wave64 launch shape. */
typedef enum nus_wg_preset {
    NUS_WG_SQUARE_8X8 = 0,
    NUS_WG_SQUARE_16X16 = 1,
    NUS_WG_WIDE_32X8 = 2,
    NUS_WG_TALL_8X32 = 3
} nus_wg_preset;

/* Bilinear arithmetic variant.  CPU form is the oracle; the WGSL form
 * is offered for callers that need byte-equality with the wgpu shader instead. */
typedef enum nus_bilinear_variant {
    NUS_BILINEAR_CPU = 0, /* Nu_scale/src/upscale/common.rs:199-231 */
    NUS_BILINEAR_WGSL = 1 /* nu_scaler_core/src/upscale/mod.rs:209-263 */
} nus_bilinear_variant;

/* Lanczos accumulate mode. FMA: fused multiply-add (within +-1 LSB of the oracle);
 * EXACT: separate multiply and add, same rounding sequence as the CPU restatement. */
typedef enum nus_lanczos_mode {
    NUS_LANCZOS_FMA = 0,
    NUS_LANCZOS_EXACT = 1
} nus_lanczos_mode;

typedef struct nus_upscaler nus_upscaler;
typedef struct nus_interp nus_interp;

/* ---- library ------------------------------------------------------------------ */

int nus_abi_version(void);
/* Number of usable HIP devices (0 when none; never fails). */
int nus_device_count(void);
/* HBM of one device in bytes (hipMemGetInfo): what PyAdvancedWgpuUpscaler.get_vram_stats reports
 * (nu_scaler_core/src/lib.rs:539-584, gpu/memory.rs:731-764).  NUS_ERR_NO_DEVICE without a device. */
int nus_device_memory_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/* Host buffers the caller re-uses (a frame pool, the Vec a Rust caller keeps between calls) can be pinned once: the host entry
 * points (nus_upscaler_upscale, _upscale_batch, nus_interp_interpolate) then DMA straight from / into them and skip the copy
 * through the library's pinned staging -- upscale_batch: 0.65 instead of 0.85 ms per 1080p -> 4K frame, 0.92 of the box's D2H
 * ceiling.  hipHostRegister / hipHostUnregister underneath (page-aligned pieces of the process's address space; the buffer
 * must not be freed or re-allocated while pinned).  Buffers that are not pinned work as before.  The library never pins a
 * caller's buffer by itself: it cannot know when the memory is given back. */
int nus_host_pin(void *buffer, size_t bytes);
/* Only a pointer nus_host_pin accepted and that has not been unpinned since (anything else: NUS_ERR_INVALID_ARGUMENT, the runtime
 * is not asked).  A failing hipHostUnregister is NUS_ERR_HIP and the range stays in the library's record (nus_host_ranges): the
 * registration may still exist, and a registration that outlives its buffer is exactly what must not go unnoticed. */
int nus_host_unpin(void *buffer);
/* The road between HBM and a caller's host buffer for users of the *_device entry points -- what the reference's upscale() does at
 * its end (map the staging buffer, wait, to_vec: nu_scaler_core/src/upscale/mod.rs:1041-1057) and before its dispatch
 * (queue.write_buffer, :968-1008).  The host pointer may be pageable: it is NEVER handed to the HIP runtime (whose pageable copies
 * pin the caller's pages on the fly and cache that registration by address -- a freed and re-used heap block then meets a stale
 * one; docs/d2h_fault_analysis.md).  The bytes travel through a ring of pinned chunks the library allocates on first use
 * (4 x 8 MiB per device, kept for the life of the process) and are moved between ring and buffer by the copy threads of the host
 * path while the next chunk is on the wire.  A pinned host buffer (hipHostMalloc, nus_host_pin) is copied to / from directly.
 *   nus_download: ordered after the work already enqueued on `stream`; returns when host_dst holds the bytes.
 *   nus_upload:   returns when host_src may be re-used; the device bytes are in place for work enqueued on `stream` afterwards.
 * `stream`: a hipStream_t of the device that owns the device pointer (NULL: its null stream).  Concurrent transfers on one device
 * take turns.  NUS_ERR_INVALID_ARGUMENT for null pointers or a device pointer the runtime does not know as device memory. */
int nus_download(void *host_dst, const void *d_src, size_t bytes, void *stream);
int nus_upload(void *d_dst, const void *host_src, size_t bytes, void *stream);
/* Diagnostics (not in the reference).  The host ranges the library has registered with the runtime (nus_host_pin), allocated as
 * pinned memory itself (slots, staging, the transfer ring) or left a transparent-huge-page hint on (fresh result buffers of the
 * host path).  history = 0: the live entries; 1: the last <= 128 events, oldest first (op 1 = added, 0 = removed).
 * Returns the number of records written (<= cap). */
typedef struct nus_host_range {
    uint64_t seq;  /* order of the event, process-wide, from 1 */
    uintptr_t lo;  /* first byte */
    uintptr_t hi;  /* one past the last byte */
    uint32_t kind; /* 1 pinned by the caller (nus_host_pin), 2 hipHostMalloc by the library, 3 MADV_HUGEPAGE hint */
    uint32_t op;
} nus_host_range;
size_t nus_host_ranges(nus_host_range *out, size_t cap, int history);
/* On SIGABRT / SIGSEGV / SIGBUS / SIGILL / SIGFPE write, to a duplicate of descriptor `fd` and with async-signal-safe calls only:
 * the native backtrace of the raising thread (which library called abort()), the ranges above, and /proc/self/maps -- then hand
 * over to the handler that was installed before (or the default action).  Off unless called; idempotent.  A GPU page fault
 * reaches a process as ROCr's abort() with an address in its message: this is what lets that address be placed afterwards. */
int nus_install_fatal_trace(int fd);
/* Calibration of the box a measurement runs on (not in the reference; bench.py's denominators next to the 8 TB/s spec figure --
 * SURVEY.md section 8(d) "on-box copy ceiling" -- never on the product path).  Enqueues ONE plain kernel (or the runtime's
 * copy) on `stream`; the caller brackets it with events.  kind: 0 hipMemcpyDtoDAsync of `bytes`; 1 stream copy, 16 B per lane,
 * read one write one; 2 write-only stream of `bytes` into d_dst; 3 read-only stream of `bytes` from d_src (d_dst: >= 4 bytes of
 * device memory); 4 one read : four writes -- `bytes` read from d_src, 4 * `bytes` written to d_dst, each store instruction one
 * contiguous KiB per wave (the byte mix of a x2 upscale with no arithmetic); 5 VGPR-only f32 FMA chains, nothing touches
 * memory: 2048 blocks x 256 lanes x 16 chains x `iters` FMAs (d_dst: >= 2 MiB, never written).  Pointers 16-byte aligned,
 * `bytes` a multiple of 16 (kind 4: of 1024 -- whole waves, each of which writes four contiguous KiB; anything else is refused). */
int nus_probe_device(int kind, const void *d_src, void *d_dst, size_t bytes, uint32_t iters, void *stream);
/* Pieces (copies and page-populate requests) of the host path's helper-thread pool that are still queued or running, process-wide
 * (not in the reference; diagnostics).  0 whenever no host call is in progress: every entry point waits for what it queued
 * before it returns, on its error paths too -- the pieces point into the caller's buffers. */
size_t nus_host_pending_pieces(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char *nus_last_error(void);
const char *nus_status_string(int status);

/* ---- Upscaler (trait Upscaler, mod.rs:67-88) ----------------------------------- */

/* WgpuUpscaler::new(quality, algorithm).  Never touches the GPU. NULL on bad enum. */
nus_upscaler *nus_upscaler_create(int algorithm, int quality);
/* UpscalerFactory::create_upscaler(technology, quality) (mod.rs:95-117):
 * Wgpu -> bilinear; FSR / DLSS / None / Fallback -> nearest. */
nus_upscaler *nus_upscaler_create_for_technology(int technology, int quality);
void nus_upscaler_destroy(nus_upscaler *h);

/* Select the HIP device (default 0).  Must precede initialize. */
int nus_upscaler_set_device(nus_upscaler *h, int device);
int nus_upscaler_set_bilinear_variant(nus_upscaler *h, int variant);
int nus_upscaler_set_lanczos_mode(nus_upscaler *h, int mode);
/* Tuning / test knobs: "force_general" (0/1, before initialize: never pick an x2
 * fast path), "force_per_pixel" (0/1, before initialize: resize without the LDS row kernel),
 * "force_rows" (0/1, before initialize: resize without the register-window variant of it),
 * "rows_per_wave" (fixed-factor resize kernels: input rows per wave, 0 = auto), "unit_order" (below),
 * "down_seg_width" (0..64, before initialize: output columns per wave of the down-scaling kernel, 0 = auto),
 * "pq_narrow" (1 default / 0: see nus_upscaler_get_option),
 * host path: "single_bands" (1 = nus_upscaler_upscale sends one frame through the pipeline in row bands where the
 * kernel allows it, default; 0 = whole), "single_out_plan" (0..3) / "batch_out_chunks" (1..8): how a pageable
 * output frame is cut into device-to-host pieces when it is alone in the pipeline / has others behind it. */
int nus_upscaler_set_option(nus_upscaler *h, const char *key, int64_t value);
/* What the library decided (diagnostics; not in the reference): "pq_p" / "pq_q" (the factor P / Q of the small-rational-factor
 * resize kernel, 0 when another kernel runs), "pq_narrow_active" (1: that kernel sums the 4 non-zero taps of a support-2 filter,
 * option "pq_narrow" 0 turns it off: same bytes either way), "rows_per_wave".  Unknown key: NUS_ERR_INVALID_ARGUMENT. */
int nus_upscaler_get_option(nus_upscaler *h, const char *key, int64_t *value);
/* Channel order of the input frames.  Captured frames arrive as BGRA and the reference swizzles them
 * on the CPU before upscaling (nu_scaler_core/src/lib.rs:251-270); with NUS_FORMAT_BGRA8 the kernels
 * do it inside their loads (one v_perm_b32 per loaded pixel, no extra pass).  Output is always RGBA8.
 * May be called at any time. */
typedef enum nus_pixel_format {
    NUS_FORMAT_RGBA8 = 0,
    NUS_FORMAT_BGRA8 = 1,
    /* alpha byte undefined (BGRX / XRGB capture surfaces): read as 255, so the output alpha is 255 */
    NUS_FORMAT_RGBX8 = 2,
    NUS_FORMAT_BGRX8 = 3
} nus_pixel_format;
int nus_upscaler_set_input_format(nus_upscaler *h, int format);
/* FSR1-style passes: the `sharpness` uniform of each shader (fsr.rs:35, :178), <= 1.  A negative
 * value keeps the default: EASU 0 (build-defined; the reference never assigns it), RCAS by quality
 * as the reference's CPU FSR path does -- Ultra 0.8, Quality 0.7, Balanced 0.6, else 0.5
 * (Nu_scale/src/upscale/fsr3.rs:231-236).  May be called at any time. */
int nus_upscaler_set_sharpness(nus_upscaler *h, float easu, float rcas);
int nus_upscaler_get_sharpness(const nus_upscaler *h, float *easu, float *rcas);

/* Upscaler::initialize (mod.rs:875-933).  Builds the per-axis tables on the host,
 * allocates device + pinned staging buffers.  Re-initialising with new dimensions
 * is allowed. */
int nus_upscaler_initialize(nus_upscaler *h, uint32_t in_w, uint32_t in_h,
                            uint32_t out_w, uint32_t out_h);

/* Upscaler::upscale (mod.rs:935-1058): host frame in, host frame out.
 * in_len must equal in_w*in_h*4; out_cap must be >= out_w*out_h*4. */
int nus_upscaler_upscale(nus_upscaler *h, const uint8_t *in, size_t in_len,
                         uint8_t *out, size_t out_cap);
/* WgpuUpscaler::upscale_batch (mod.rs:609-640): n host frames, pipelined over
 * H2D / kernel / D2H streams instead of a rayon map. */
int nus_upscaler_upscale_batch(nus_upscaler *h, const uint8_t *const *ins,
                               const size_t *in_lens, size_t n,
                               uint8_t *const *outs, size_t out_cap_each);

/* A persistent ring over the same three pipeline slots, for callers that get their frames one at a time (a capture loop: the
 * legacy app's FrameBuffer feeding upscale(), Nu_scale/src/capture/frame_buffer.rs:11-50, Nu_scale/src/lib.rs:107-190):
 *   stream_open    starts the ring (NUS_ERR_NOT_INITIALIZED before initialize; one stream per upscaler);
 *   stream_submit  stages and enqueues one frame and returns -- it blocks only while 3 frames are in flight -- with a
 *                  ticket (0, 1, 2 ...); `out` must stay valid until that ticket has been waited for (or the stream closed);
 *   stream_wait    returns when the frame with that ticket is in its `out` buffer (frames complete in submission order; may
 *                  be called from another thread than stream_submit);
 *   stream_close   waits for everything in flight and stops the ring (also done by initialize and destroy).
 * While a stream is open, nus_upscaler_upscale / _upscale_batch on the same handle fail (the slots are in use).  H2D of frame
 * i+1, the kernel of frame i and the D2H / copy-out of frame i-1 overlap exactly as inside nus_upscaler_upscale_batch. */
int nus_upscaler_stream_open(nus_upscaler *h);
int nus_upscaler_stream_submit(nus_upscaler *h, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                               uint64_t *ticket);
int nus_upscaler_stream_wait(nus_upscaler *h, uint64_t ticket);
int nus_upscaler_stream_close(nus_upscaler *h);

/* Device-resident path: d_in holds n_frames contiguous input frames already in HBM,
 * d_out receives n_frames contiguous output frames.  Enqueued on `stream`
 * (a hipStream_t, NULL = default stream); does not synchronise. */
int nus_upscaler_upscale_device(nus_upscaler *h, const void *d_in, void *d_out,
                                uint32_t n_frames, void *stream);

/* Fused "interpolate with zero flow, then upscale the in-between frame" -- the GUI's own sequence
 * (nu_scaler_py/nu_scaler/main.py:999-1008) -- without materialising the 1080p in-between frame:
 * unit i blends frames d_a + i*a_stride and d_b + i*b_stride at t (per channel
 * trunc((1-t) a + t b), the pixel nus_interp_interpolate would have produced) inside the resize
 * kernel's row loads.  Available for the exact-x2 resize kernels (Lanczos-3 / bicubic / triangle);
 * otherwise NUS_ERR_UNSUPPORTED and the caller runs the two stages separately.  Results equal
 * nus_interp_interpolate_device followed by nus_upscaler_upscale_device bit for bit. */
int nus_upscaler_upscale_blend_device(nus_upscaler *h, const void *d_a, size_t a_stride, const void *d_b,
                                      size_t b_stride, float t, void *d_out, uint32_t n_frames, void *stream);

/* The whole step of the GUI's frame loop for a batch of pairs in ONE launch (nu_scaler_py/nu_scaler/main.py:999-1008 interpolates
 * and then upscales; the BASELINE metric upscales the real frame too): for unit i, with A_i = d_a + i*a_stride and
 * B_i = d_b + i*b_stride (stride 0 = tightly packed frames),
 *   d_out_real[i] = upscale(A_i)                       what nus_upscaler_upscale_device writes,
 *   d_mid[i]      = blend(A_i, B_i) at t, zero flow    what nus_interp_interpolate_device writes (d_mid may be NULL),
 *   d_out_mid[i]  = upscale(blend(A_i, B_i))           what nus_upscaler_upscale_blend_device writes,
 * bit for bit.  Every (frame, row block, strip) is walked by two waves of the same kernel side by side -- one on A_i alone, one
 * blending A_i and B_i on load -- so each input row reaches HBM's bus about once per step instead of four times and the
 * in-between frame is written, never re-read.  Exact-x2 resize kernels (Lanczos-3 / bicubic / triangle) only; otherwise
 * NUS_ERR_UNSUPPORTED and the caller runs the three stages separately.  Enqueue only, no allocation.
 * Option "unit_order" (nus_upscaler_set_option): 1 = row-block-major wave order (default), 0 = frame-major. */
int nus_upscaler_upscale_unit_device(nus_upscaler *h, const void *d_a, size_t a_stride, const void *d_b, size_t b_stride,
                                     float t, void *d_mid, void *d_out_real, void *d_out_mid, uint32_t n_units,
                                     void *stream);

const char *nus_upscaler_name(const nus_upscaler *h); /* "WgpuNearestUpscaler" / "WgpuBilinearUpscaler" (mod.rs:1060-1066) / "Hip{Lanczos3,Bicubic,Triangle}Upscaler" */
int nus_upscaler_algorithm(const nus_upscaler *h);
int nus_upscaler_quality(const nus_upscaler *h);
int nus_upscaler_set_quality(nus_upscaler *h, int quality);
int nus_upscaler_is_initialized(const nus_upscaler *h);
size_t nus_upscaler_input_size(const nus_upscaler *h);  /* in_w*in_h*4, 0 before initialize */
size_t nus_upscaler_output_size(const nus_upscaler *h); /* out_w*out_h*4 */
const char *nus_upscaler_last_error(const nus_upscaler *h);
/* Kernel time (hipEvent pair) of the last host-path upscale; NUS_ERR_NOT_INITIALIZED if none. */
int nus_upscaler_last_gpu_ms(const nus_upscaler *h, double *ms_out);
/* Device-path kernel timing.  With profiling enabled every nus_upscaler_upscale_device call
 * brackets its main kernel launch with a hipEvent pair on the caller's stream (the Lanczos
 * edge-column pass is outside the bracket); profile_collect waits for the recorded pairs,
 * returns the number of launches and their summed duration, and resets the counters. */
int nus_upscaler_set_profiling(nus_upscaler *h, int enabled);
int nus_upscaler_profile_collect(nus_upscaler *h, uint64_t *launches, double *total_ms);
/* Name of the kernel variant chosen at initialize (e.g. "lanczos3_x2_regwin"). */
const char *nus_upscaler_kernel_variant(const nus_upscaler *h);

/* Shared LUT exchange for multi-GPU runs: rank 0 exports the per-axis tables built at
 * initialize, the blob is broadcast (RCCL), other ranks import it so every GPU uses
 * bit-identical weights.  export returns the blob size (or negative status);
 * with buf == NULL it only reports the size. */
int64_t nus_upscaler_export_tables(const nus_upscaler *h, void *buf, size_t cap);
int nus_upscaler_import_tables(nus_upscaler *h, const void *buf, size_t len);

/* Host-only (no GPU needed): build the same blob directly from the dimensions, and
 * check a received blob against them.  variant: nus_bilinear_variant. */
int64_t nus_tables_build_blob(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                              int variant, void *buf, size_t cap);
int64_t nus_tables_build_blob_for(int algorithm, uint32_t in_w, uint32_t in_h, uint32_t out_w,
                                  uint32_t out_h, int variant, void *buf, size_t cap);
int nus_tables_validate_blob(const void *buf, size_t len, uint32_t in_w, uint32_t in_h,
                             uint32_t out_w, uint32_t out_h);

/* ---- host-only table builders (no GPU needed; used by initialize) --------------- */

#define NUS_RESIZE_MAX_TAPS 32

/* Lanczos-3 tap windows of one axis, image-0.24.9 convention (half-pixel centres,
 * support 3*max(in/out,1), weights normalised to sum 1).  left/ntaps: out_n entries;
 * weights: out_n * NUS_RESIZE_MAX_TAPS floats, zero padded.
 * Returns the largest ntaps or a negative status. */
int nus_lanczos3_build_axis(uint32_t in_n, uint32_t out_n, int32_t *left,
                            uint32_t *ntaps, float *weights);
/* Same for filter 0 = Lanczos3, 1 = CatmullRom, 2 = Triangle. */
int nus_resize_build_axis(int filter, uint32_t in_n, uint32_t out_n, int32_t *left,
                          uint32_t *ntaps, float *weights);
/* Nearest source index per output index: min(o*in_n/out_n, in_n-1). */
int nus_nearest_build_axis(uint32_t in_n, uint32_t out_n, uint32_t *src);
/* Bilinear i0 / frac per output index. variant: nus_bilinear_variant. */
int nus_bilinear_build_axis(uint32_t in_n, uint32_t out_n, int variant,
                            uint32_t *i0, float *frac);

/* ---- Frame interpolator (WgpuFrameInterpolator) -------------------------------- */

/* WgpuFrameInterpolator::new_py(preset) (wgpu_interpolator.rs:172-212). */
nus_interp *nus_interp_create(int wg_preset);
void nus_interp_destroy(nus_interp *h);
int nus_interp_set_device(nus_interp *h, int device);
/* Channel order of BOTH input frames (see nus_upscaler_set_input_format); the new frame is RGBA8. */
int nus_interp_set_input_format(nus_interp *h, int format);
/* Element type of the flow field nus_interp_interpolate_device reads: NUS_FLOW_F32 (default, 2 x f32 per pixel,
 * the Rg32Float layout of wgpu_interpolator.rs:1211, :1418) or NUS_FLOW_F16 (2 x IEEE half per pixel, the
 * Rg16Float texture the reference's live path binds, wgpu_interpolator.rs:276: half the flow bytes).  Each
 * half is widened to f32 exactly and the arithmetic is the same.  The host entry point always takes f32. */
typedef enum nus_flow_format {
    NUS_FLOW_F32 = 0,
    NUS_FLOW_F16 = 1
} nus_flow_format;
int nus_interp_set_flow_format(nus_interp *h, int format);
/* Arithmetic of the dense-flow warp (flow != NULL), as nus_upscaler_set_lanczos_mode for the resize filters:
 *   NUS_INTERP_MODE_EXACT (default)  every product and sum rounded separately, as the CPU code of
 *                                    interpolation/mod.rs:467-510 and :386-411 -- bit-exact against the oracle;
 *   NUS_INTERP_MODE_FMA              each bilinear lerp as a + f (b - a) with one fused multiply-add (the blend of the two
 *                                    truncated samples keeps the CPU's roundings): within 1 LSB of EXACT, fewer than 0.1 % of
 *                                    the samples differ, 0.8x the instructions.
 * The zero-flow path (the reference's live behaviour) is the same exact streaming blend in both modes. */
typedef enum nus_interp_mode_t { NUS_INTERP_MODE_EXACT = 0, NUS_INTERP_MODE_FMA = 1 } nus_interp_mode_t;
int nus_interp_set_mode(nus_interp *h, int mode);
int nus_interp_mode(const nus_interp *h);

/* trait FrameInterpolator (nu_scaler_core/src/interpolation/mod.rs:29-44) -- the shape the dead-code trait gives an
 * interpolator, next to the live pyclass's per-call form below:
 *   initialize(width, height)          :31  buffers for that frame size up front; same size again = no-op (:306-308)
 *   interpolate(frame1, frame2, t)     :34  frames of the initialize() size, zero flow; before initialize:
 *                                           NUS_ERR_NOT_INITIALIZED, "Interpolator not initialized" (:368-370)
 *   name()                             :37
 *   set_quality(q) / quality()         :40-43  nus_interp_quality_level; kept and reported, arithmetic unchanged */
typedef enum nus_interp_quality_level {
    NUS_INTERP_QUALITY_HIGH = 0, /* InterpolationQuality::High   (interpolation/mod.rs:8-14) */
    NUS_INTERP_QUALITY_MEDIUM = 1,
    NUS_INTERP_QUALITY_LOW = 2
} nus_interp_quality_level;
int nus_interp_initialize(nus_interp *h, uint32_t width, uint32_t height);
int nus_interp_interpolate_frames(nus_interp *h, const uint8_t *frame1, size_t len1, const uint8_t *frame2, size_t len2,
                                  float t, uint8_t *out, size_t out_cap);
const char *nus_interp_name(const nus_interp *h);
int nus_interp_set_quality(nus_interp *h, int quality);
int nus_interp_quality(const nus_interp *h);

/* interpolate_py (wgpu_interpolator.rs:215-491): host frames in, host frame out.
 * flow == NULL -> zero flow (the live reference behaviour); otherwise w*h*2 floats
 * (dx, dy) = pixel delta from frame A to frame B (warp_blend.wgsl:29-37).
 * a_len and b_len must equal w*h*4 (else NUS_ERR_SIZE_MISMATCH with the text of
 * wgpu_interpolator.rs:234-237); out_cap >= w*h*4. */
int nus_interp_interpolate(nus_interp *h, const uint8_t *a, size_t a_len,
                           const uint8_t *b, size_t b_len, const float *flow,
                           uint32_t w, uint32_t hgt, float t,
                           uint8_t *out, size_t out_cap);

/* Device-resident path: n_pairs independent pairs; pair i reads
 * d_a + i*a_stride, d_b + i*b_stride (byte strides; a sliding stream uses
 * d_b = d_a + frame_bytes with both strides = frame_bytes), optional
 * d_flow + i*w*h*8 (NUS_FLOW_F32; i*w*h*4 and 4-byte alignment with NUS_FLOW_F16), writes d_out + i*w*h*4.
 * Enqueued on `stream`, no sync. */
int nus_interp_interpolate_device(nus_interp *h, const void *d_a, size_t a_stride,
                                  const void *d_b, size_t b_stride,
                                  const void *d_flow, uint32_t w, uint32_t hgt,
                                  float t, void *d_out, uint32_t n_pairs, void *stream);

/* get_last_gpu_duration_ms: NUS_OK and *ms_out set, or NUS_ERR_NOT_INITIALIZED when
 * no interpolation has run yet (the reference returns None). */
int nus_interp_last_gpu_ms(const nus_interp *h, double *ms_out);
const char *nus_interp_last_error(const nus_interp *h);

/* ---- Frame queue + capture swizzle ("next" row: SURVEY.md section 8f rank 3) -----------
 * nus_frame_queue mirrors the legacy FrameBuffer (Nu_scale/src/capture/frame_buffer.rs:11-100):
 * bounded, drop-oldest on overflow, consumers read the latest frame.  Host memory only. */
typedef struct nus_frame_queue nus_frame_queue;

nus_frame_queue *nus_frame_queue_create(size_t capacity); /* reference default: 5 */
void nus_frame_queue_destroy(nus_frame_queue *q);
/* Copies the frame in; returns the total number of frames dropped so far (>= 0) or a status. */
int64_t nus_frame_queue_add(nus_frame_queue *q, const uint8_t *rgba, uint32_t w, uint32_t hgt);
/* Latest frame (stays queued) / oldest frame (removed).  Waits up to timeout_ms when empty.
 * Returns 1 and fills out (cap >= w*h*4), 0 when no frame arrived, negative status on error. */
int nus_frame_queue_latest(nus_frame_queue *q, int64_t timeout_ms, uint8_t *out, size_t out_cap,
                           uint32_t *w, uint32_t *hgt, uint64_t *sequence);
int nus_frame_queue_pop(nus_frame_queue *q, int64_t timeout_ms, uint8_t *out, size_t out_cap,
                        uint32_t *w, uint32_t *hgt, uint64_t *sequence);
size_t nus_frame_queue_size(const nus_frame_queue *q);
size_t nus_frame_queue_capacity(const nus_frame_queue *q);
uint64_t nus_frame_queue_dropped(const nus_frame_queue *q);

/* BGRA -> RGBA of a captured frame on the GPU (the CPU loop of nu_scaler_core/src/lib.rs:251-270).
 * Device pointers, n_pixels pixels, in place allowed; enqueues on `stream`. */
int nus_swizzle_bgra_to_rgba_device(const void *d_in, void *d_out, size_t n_pixels, void *stream);

/* ---- Optical-flow front end ("next" row: SURVEY.md section 8f rank 1) --------------
 * Mirrors WgpuFrameInterpolator::build_pyramid / compute_coarse_flow
 * (nu_scaler_core/src/wgpu_interpolator.rs:969-1203) and the shaders
 * gaussian_blur_{h,v}.wgsl, downsample.wgsl, horn_schunck.wgsl, flow_upsample.wgsl.
 * Images: f32 RGBA (4 floats / pixel, u8/255); flows: 2 floats / pixel (dx, dy), the
 * pixel delta from frame A to frame B that nus_interp_interpolate consumes. */
typedef struct nus_flow nus_flow;

nus_flow *nus_flow_create(void);
void nus_flow_destroy(nus_flow *h);
int nus_flow_set_device(nus_flow *h, int device);
/* 1 (default): derivatives once per level and several Jacobi steps per launch, the kernel chosen by
 * the size of the batch (2: always the LDS-tile kernel, 3: always the register-pipelined one);
 * 0: one plain kernel per step, the shader's structure.  Results are bit-identical. */
int nus_flow_set_tiled(nus_flow *h, int enabled);
/* Arithmetic of the estimators' Horn-Schunck steps, as nus_upscaler_set_lanczos_mode / nus_interp_set_mode have it for their
 * kernels.  NUS_FLOW_EXACT (default): every stage bit-identical to the oracle's restatement of the shaders.  NUS_FLOW_FAST: the
 * Jacobi steps with separable 3x3 sums, a multiply by 1/9, a precomputed reciprocal of lambda + Ix^2 + Iy^2 and FMAs
 * (shaders/horn_schunck.wgsl:24-92 in its cheapest f32 form; every level in the register-pipelined kernel) -- the flow within
 * 1e-3 px of the exact one, the frame interpolated with it within 1 LSB on < 0.1 % of its samples; 2.2x fewer instructions per
 * step.  The reference never runs this front end (wgpu_interpolator.rs:1156-1203 is unwired), so neither mode has a fixture.
 * The primitives below are always exact. */
#define NUS_FLOW_EXACT 0
#define NUS_FLOW_FAST 1
int nus_flow_set_mode(nus_flow *h, int mode);
int nus_flow_mode(const nus_flow *h);
const char *nus_flow_last_error(const nus_flow *h);

/* primitives on host buffers */
int nus_flow_rgba8_to_f32(nus_flow *h, const uint8_t *in, uint32_t w, uint32_t hgt, float *out);
int nus_flow_blur(nus_flow *h, const float *in, uint32_t w, uint32_t hgt, float *out); /* H pass, then V pass */
int nus_flow_downsample(nus_flow *h, const float *in, uint32_t w, uint32_t hgt, float *out); /* -> (w+1)/2 x (h+1)/2 */
/* `iterations` Jacobi steps; flow_in == NULL starts from zero (compute_coarse_flow). */
int nus_flow_horn_schunck(nus_flow *h, const float *i1, const float *i2, const float *flow_in,
                          uint32_t w, uint32_t hgt, float lambda, uint32_t iterations, float *flow_out);
int nus_flow_upsample(nus_flow *h, const float *src, uint32_t sw, uint32_t sh,
                      float *dst, uint32_t dw, uint32_t dh, float scale);

/* Full estimator: pyramids of both RGBA8 frames (`levels`), `coarse_iters` steps from zero
 * flow at the coarsest level, then per finer level a x2 upsample (vectors x2) and
 * `refine_iters` steps.  flow_out: w*h*2 floats. */
int nus_flow_estimate(nus_flow *h, const uint8_t *a, const uint8_t *b, uint32_t w, uint32_t hgt,
                      uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda,
                      float *flow_out);
/* Device-resident variant; enqueues on `stream`, no synchronisation. */
int nus_flow_estimate_device(nus_flow *h, const void *d_a, const void *d_b, uint32_t w, uint32_t hgt,
                             uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda,
                             void *d_flow_out, void *stream);
/* Flows between consecutive frames of a device-resident stream (what the interpolator's batch entry
 * point consumes): d_frames = n_frames RGBA8 frames back to back, d_flows = n_frames - 1 flows,
 * flow k = frame k -> frame k+1.  Same result as n_frames - 1 calls of nus_flow_estimate_device;
 * each frame's pyramid is built once.  Enqueues on `stream`, no synchronisation. */
int nus_flow_estimate_device_stream(nus_flow *h, const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t hgt,
                                    uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda,
                                    void *d_flows, void *stream);
/* Motion-compensated interpolation of a device-resident stream as ONE pipeline -- the reference's intended interpolate()
 * (wgpu_interpolator.rs:881-935: build_pyramid -> compute_coarse_flow -> warp): the n_frames - 1 flows of
 * nus_flow_estimate_device_stream AND the in-between frame of every pair (k, k + 1) at time_t in [0, 1], warped + blended with that
 * flow (warp_blend.wgsl:25-43; dense-flow warp in FMA mode: within 1 LSB of the CPU blend, as nus_interp_set_mode(.., FMA)), tightly
 * packed RGBA8 at d_mid.  d_flows may be NULL (the in-between frames only: the flows then stay in the estimator's workspace).
 * flow_format: NUS_FLOW_F32 -- same bytes as nus_flow_estimate_device_stream followed by nus_interp_interpolate_device in FMA mode;
 * NUS_FLOW_F16 -- the flows between estimator and warp (and at d_flows: w*h*4 bytes per pair) are the reference's live layout,
 * Rg16Float (wgpu_interpolator.rs:276): each the f32 flow rounded to nearest even, and the warp reads them as such (2^-11 relative
 * resolution: 5e-4 px on a 1-2 px flow; half the bytes of the hand-off).  Pointers 16-byte aligned. */
int nus_flow_interpolate_device_stream(nus_flow *h, const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t hgt,
                                       uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda, float time_t,
                                       int flow_format, void *d_flows, void *d_mid, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NUSCALER_HIP_H */

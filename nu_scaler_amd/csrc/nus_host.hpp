// nus_host.hpp -- C++ host side above the kernels, mirroring the reference's Rust
// interfaces for this path (the Rust toolchain is absent in the build image, so the
// host layer the north star asks for in Rust is written in C++; INTEGRATION.md shows
// the Rust shim that binds the C ABI on top of it).
//
//   nus::Upscaler          <-> trait Upscaler            nu_scaler_core/src/upscale/mod.rs:67-88
//   nus::HipUpscaler       <-> WgpuUpscaler              nu_scaler_core/src/upscale/mod.rs:266-1086
//   nus::UpscalerFactory   <-> UpscalerFactory           nu_scaler_core/src/upscale/mod.rs:91-117
//   nus::FrameInterpolator <-> trait FrameInterpolator   nu_scaler_core/src/interpolation/mod.rs:29-44
//   nus::HipFrameInterpolator <-> WgpuFrameInterpolator  nu_scaler_core/src/wgpu_interpolator.rs:130-498
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "nus_kernels.hpp"
#include "nus_copy.hpp"
#include "nus_tables.hpp"

namespace nus {

// nus_status values (kept in sync with include/nuscaler_hip.h).
enum Status : int {
    kOk = 0,
    kInvalidArgument = -1,
    kNotInitialized = -2,
    kSizeMismatch = -3,
    kHipError = -4,
    kNoDevice = -5,
    kUnsupported = -6,
    kOutOfMemory = -7,
};

enum class Algorithm : int { Nearest = 0, Bilinear = 1, Lanczos3 = 2, Bicubic = 3, Triangle = 4, Fsr1 = 5, FsrEasu = 6, FsrRcas = 7 };
enum class Quality : int { UltraPerformance = 0, Ultra, Quality, Balanced, Performance, Native };
enum class Technology : int { None = 0, FSR, DLSS, Wgpu, Fallback };

void set_thread_error(const std::string &msg);
const char *thread_error();

// trait Upscaler (upscale/mod.rs:67-88).  Result<()> becomes a Status + last_error().
class Upscaler {
public:
    virtual ~Upscaler() = default;
    virtual int initialize(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h) = 0;
    virtual int upscale(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap) = 0;
    virtual const char *name() const = 0;
    virtual Quality quality() const = 0;
    virtual int set_quality(Quality q) = 0;
    virtual const char *last_error() const = 0;
};

class HipUpscaler final : public Upscaler {
public:
    HipUpscaler(Quality q, Algorithm a);
    ~HipUpscaler() override;
    HipUpscaler(const HipUpscaler &) = delete;
    HipUpscaler &operator=(const HipUpscaler &) = delete;

    int initialize(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h) override;
    int upscale(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap) override;
    const char *name() const override;
    Quality quality() const override { return quality_; }
    int set_quality(Quality q) override;
    const char *last_error() const override { return error_.c_str(); }

    // upscale_batch (upscale/mod.rs:609-640): H2D / kernel / D2H pipelined over slots.
    int upscale_batch(const uint8_t *const *ins, const size_t *in_lens, size_t n, uint8_t *const *outs,
                      size_t out_cap_each);
    // Persistent ring over the same slots: frames go in one at a time (stream_submit blocks only while kSlots frames are in
    // flight), come out in submission order, and stream_wait(ticket) returns once frame `ticket` is in its output buffer.
    // The capture loop's shape (Nu_scale/src/capture/frame_buffer.rs:11-50 -> upscale); while a stream is open the other host
    // entry points of this upscaler are refused.  stream_wait may be called from another thread than stream_submit.
    int stream_open();
    int stream_submit(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, uint64_t *ticket);
    int stream_wait(uint64_t ticket);
    int stream_close();
    // frames already resident in HBM; enqueue only.
    int upscale_device(const void *d_in, void *d_out, uint32_t n_frames, hipStream_t stream);
    // Fused "interpolate (zero flow) then upscale the in-between frame": unit i reads frame
    // d_a + i*a_stride and d_b + i*b_stride.  x2 resize kernels only (else kUnsupported).
    int upscale_blend_device(const void *d_a, size_t a_stride, const void *d_b, size_t b_stride, float t, void *d_out,
                             uint32_t n_frames, hipStream_t stream);

    // One whole pipeline step in one launch of the x2 resize kernel: for every pair (A_i, B_i) the up-scaled real frame A_i
    // (d_out_real), the zero-flow in-between frame at t (d_mid, input-size frames; may be null) and the up-scaled in-between
    // frame (d_out_mid).  The same bytes as interpolate_device + 2 x upscale_device; x2 resize kernels only.
    int upscale_unit_device(const void *d_a, size_t a_stride, const void *d_b, size_t b_stride, float t, void *d_mid,
                            void *d_out_real, void *d_out_mid, uint32_t n_units, hipStream_t stream);

    int set_device(int device);
    int set_bilinear_variant(int variant);
    int set_lanczos_mode(int mode);
    int set_option(const char *key, int64_t value);
    int get_option(const char *key, int64_t *value);
    // nus_pixel_format: 0 RGBA8 (default), 1 BGRA8, 2 RGBX8, 3 BGRX8 -- swizzled / made opaque inside the kernels' loads
    // (the reference's CPU loop: nu_scaler_core/src/lib.rs:251-270).  Output is always RGBA8.
    int set_input_format(int format);
    // FSR1-style passes: negative value = keep the quality-derived default.
    int set_sharpness(float easu, float rcas);
    float easu_sharpness() const;
    float rcas_sharpness() const;

    Algorithm algorithm() const { return algorithm_; }
    bool initialized() const { return initialized_; }
    size_t input_size() const { return initialized_ ? (size_t)iw_ * ih_ * 4 : 0; }
    size_t output_size() const { return initialized_ ? (size_t)ow_ * oh_ * 4 : 0; }
    bool last_gpu_ms(double *ms) const;
    const char *kernel_variant() const { return variant_name(variant_); }

    // Device-path kernel timing: with profiling on, every upscale_device() call brackets
    // its main kernel launch with a hipEvent pair on the caller's stream;
    // profile_collect() waits for them, returns launch count + summed duration, resets.
    int set_profiling(bool on);
    int profile_collect(uint64_t *launches, double *total_ms);

    int64_t export_tables(void *buf, size_t cap) const;
    int import_tables(const void *buf, size_t len);

private:
    static constexpr int kSlots = 3; // buffer_pool_size default (upscale/mod.rs:287-289)
    // The output frame comes back in kOutChunks pieces so the host copy out of the pinned staging
    // buffer overlaps the DMA of the next piece (the reference maps, copies to a Vec and copies again:
    // upscale/mod.rs:1040-1057, lib.rs:111).
    static constexpr int kOutChunks = 8;
    // A frame that is alone in the pipeline (trait Upscaler::upscale) goes through it in kBands row bands where the kernel can
    // be launched on a row range (the exact-x2 kernels): staging copy, H2D and kernel of band b+1 run while band b's output is
    // already on its way back, so the D2H engine -- what bounds the call -- starts after a quarter of the upload instead of all
    // of it (option "single_bands": 0 turns it off).
    static constexpr int kBands = 4;
    struct Slot {
        uint8_t *d_in = nullptr, *d_out = nullptr; // HBM
        uint8_t *h_in = nullptr, *h_out = nullptr; // pinned staging
        hipEvent_t in_done = nullptr;               // the frame's H2D has landed in d_in           (copy-in stream)
        hipEvent_t k_begin = nullptr, k_end = nullptr; // around its kernel launches                 (compute stream)
        hipEvent_t chunk_done[kOutChunks] = {};     // D2H of output chunk k has landed in h_out    (copy-out stream)
        hipEvent_t out_done = nullptr;              // ... of the whole frame
        bool used = false;                          // events have been recorded at least once
        hipEvent_t band_in[kBands] = {}, band_k[kBands] = {}; // banded single frame: H2D / kernel of band b done
        int nchunks = 0;                            // pieces of the frame in flight (pageable output) ...
        size_t chunk_end[kOutChunks] = {};          // ... and where each ends (set by submit_frame, read by retire_frame)
        CopyTicket populate;                        // the caller's (pageable) output buffer being made resident meanwhile
    };
    // How a pageable output frame is cut into D2H pieces.  Every piece costs the copy engine ~17 us of dead time (the event
    // between two copies of one stream: profiles/r03_host_path_copy_timeline.txt), and buys overlap of the copy out of the
    // pinned buffer with the next piece's DMA -- which a frame that is alone in the pipeline needs (trait Upscaler::upscale)
    // and a frame with others behind it does not (their DMA is what its copy-out overlaps with).
    int single_out_plan_ = 0;   // option "single_out_plan": 0 = 8 equal pieces (rounds 1-2), 1 = 1/2 1/4 1/8 1/8, 2 = 1/2 1/4 1/8 1/16 1/16, 3 = 4 equal
    int batch_out_chunks_ = 2;  // option "batch_out_chunks": equal pieces per frame of upscale_batch / the stream ring
    void plan_chunks(Slot &s, size_t out_bytes, bool alone) const;
    int single_bands_ = 1;      // option "single_bands"
    mutable std::atomic<int> inject_retire_{0}; // option "inject_retire_error" (test hook; read on the retiring thread)
    uint32_t band_alignment(uint32_t n_frames) const; // rows a band must be a multiple of; 0 = the variant has no row-range launch
    uint32_t lanczos_x2_rows_per_wave(uint32_t n_frames, bool unit) const;
    int submit_frame_banded(Slot &s, const uint8_t *in, uint8_t *out, bool *direct, uint32_t align, int populate);
    // "one host thread + 3 streams per GPU": every H2D goes down the copy-in stream, every kernel down the compute stream,
    // every D2H down the copy-out stream, tied together per frame by the slot's events -- so the copies of consecutive frames
    // sit back to back in ONE queue per direction (0.60 ms per 4K frame on the D2H engine) instead of alternating between the
    // queues of per-slot streams (0.73 ms per frame measured that way, round 3).
    hipStream_t s_in_ = nullptr, s_k_ = nullptr, s_out_ = nullptr;
    // The edge-column pass of the exact-x2 resize kernels writes columns the main kernel does not touch (round 4), so for batches
    // it is forked onto this stream -- begin event on the caller's stream, edge pass here, join event back -- and runs beside the
    // main kernel instead of behind it as a launch of 34 single-wave blocks per frame that cannot fill the GPU (3 % of the
    // one-launch step).  Created by initialize(); enqueue() allocates nothing.  Option "edge_stream": 0 = same stream.
    hipStream_t s_edge_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    int edge_stream_ = 1;

    int fail(int status, const std::string &msg);
    int fail_hip(hipError_t e, const char *what);
    int ensure_device();
    int ensure_streams();
    int ensure_slot(Slot &s, size_t in_bytes, size_t out_bytes);
    int submit_frame(Slot &s, const uint8_t *in, uint8_t *out, bool *direct, bool alone, int populate = -1);
    int submit_frame_inner(Slot &s, const uint8_t *in, uint8_t *out, bool *direct, bool alone, int populate);
    int retire_frame(Slot &s, uint8_t *out, bool direct, std::string *err) const;
    struct Ring { // state of an open stream (stream_open .. stream_close); its own mutex, never held together with a HIP call
        struct Item {
            uint8_t *out = nullptr;
            bool direct = false;
        };
        bool open = false; // guarded by mu_
        std::thread thread;
        std::mutex m;
        std::condition_variable cv;
        uint64_t submitted = 0, retired = 0;
        bool stop = false;
        int status = 0;
        std::string error;
        Item items[kSlots];
    } ring_;
    void release_slot(Slot &s);
    void release();
    int upload_tables();
    bool is_fsr() const { return algorithm_ == Algorithm::Fsr1 || algorithm_ == Algorithm::FsrEasu || algorithm_ == Algorithm::FsrRcas; }
    bool is_resize() const { return algorithm_ == Algorithm::Lanczos3 || algorithm_ == Algorithm::Bicubic || algorithm_ == Algorithm::Triangle; }
    ResizeFilter resize_filter() const;
    void choose_variant();
    bool ratio_shape(bool bilinear) const; // tables have the shape of the fixed-ratio nearest / bilinear kernels
    void choose_resize_variant(bool x2);
    void choose_general_resize_variant();
    uint32_t widest_footprint(uint32_t segw) const;
    uint32_t widest_union(uint32_t n) const;
    struct BlendSrc {
        const uint8_t *b = nullptr;
        size_t a_stride = 0, b_stride = 0;
        float t = 0.5f;
    };
    struct UnitDst { // upscale_unit_device: the two extra outputs of the one-launch step
        uint8_t *out_mid = nullptr, *mid = nullptr;
    };
    int enqueue(const uint8_t *d_in, uint8_t *d_out, uint32_t n_frames, hipStream_t stream, const BlendSrc *blend = nullptr,
                const UnitDst *unit = nullptr, uint32_t row0 = 0, uint32_t rows = 0);

    mutable std::mutex mu_;
    Quality quality_;
    Algorithm algorithm_;
    int device_ = 0;
    bool wgsl_bilinear_ = false;
    bool lanczos_exact_ = false;
    bool fsr_fast_ = false;        // option "fsr_fast"
    bool fsr_two_pass_ = true;     // option "fsr_two_pass": 0 keeps the fused LDS tile at every size
    // Fsr1TwoPass: kFsrScratchFrames EASU images (RGBA8, output size) between the two passes, and the event that keeps a second
    // batch of this handle (on whatever stream) from writing them before the first batch's RCAS has read them
    static constexpr uint32_t kFsrScratchFrames = 4;
    uint8_t *fsr_scratch_ = nullptr;
    hipEvent_t ev_fsr_ = nullptr;
    bool fsr_pending_ = false;
    bool force_general_ = false;
    bool force_per_pixel_ = false; // resize: never use the LDS row kernel
    bool force_rows_ = false;      // resize: never use the register-window variant of it
    uint32_t resize_ncols_max_ = 0; // LDS row length of the ResizeRows variant
    uint32_t down_seg_w_ = 64;      // ResizeDown: output columns per wave (choose_resize_variant)
    uint32_t down_seg_width_ = 0;   // option "down_seg_width": 0 = chosen by cost
    std::vector<uint32_t> down_rows_; // ResizeDown: per-input-row slot weights + completion (build_down_stream_tables)
    std::vector<int32_t> down_done_;
    bool resize_small_taps_ = false;
    uint32_t xs_factor_ = 0;         // 3 / 4 when the integer-factor register-window kernel is selected
    uint32_t win_outputs_per_lane_ = 4; // register-window resize: 4, or 2 for factors below ~x1.4
    uint32_t resize_union_taps_ = 0; // widest union of the tap windows of 4 adjacent outputs (0: unused)
    uint32_t rows_per_wave_ = 0; // 0: pick from the batch size
    uint32_t unit_order_ = 1;    // wave order of the unit launch: 0 frame-major, 1 row-block-major (option "unit_order")
    float easu_sharp_ = -1.0f, rcas_sharp_ = -1.0f; // < 0: derive from quality_
    int in_format_ = 0; // nus_pixel_format
    bool initialized_ = false;
    uint32_t iw_ = 0, ih_ = 0, ow_ = 0, oh_ = 0;
    Variant variant_ = Variant::NearestTable;
    AxisTables tx_, ty_;
    std::vector<float> wy6_, wx6_;
    uint32_t pq_p_ = 0, pq_q_ = 0; // Variant::LanczosPqRegWin: the factor P / Q
    bool pq_exact_fallback_ = false; // ... and whether its EXACT mode runs on the any-scale kernel instead (Q = 5)
    bool pq_narrow_ = false;         // ... and whether every output's frame slots 0 and 5 are zero (a support-2 filter: 4-tap sums)
    bool pq_narrow_allowed_ = true;  // option "pq_narrow"
    std::vector<uint32_t> xs_cls_x_, xs_cls_y_; // x3: weight class per input index, and the classes' weights
    std::vector<float> xs_wcls_x_, xs_wcls_y_;
    DeviceTables dt_;
    std::vector<void *> table_allocs_;
    Slot slots_[kSlots];
    bool have_ms_ = false;
    double last_ms_ = 0.0;
    bool profiling_ = false;
    std::vector<hipEvent_t> prof_events_; // begin/end pairs
    size_t prof_used_ = 0;                // events handed out since the last collect
    std::string error_;
};

// UpscalerFactory::create_upscaler (upscale/mod.rs:95-117).
struct UpscalerFactory {
    static std::unique_ptr<HipUpscaler> create_upscaler(Technology tech, Quality q);
};

// InterpolationQuality (interpolation/mod.rs:6-15).
enum class InterpolationQuality : int { High = 0, Medium = 1, Low = 2 };

// trait FrameInterpolator (interpolation/mod.rs:29-44): initialize(width, height), interpolate(frame1, frame2, t),
// name, set_quality, quality.  `interpolate` keeps the live pyclass's argument list (dimensions per call, optional
// flow: wgpu_interpolator.rs:215-225), of which the trait's is the special case "dimensions from initialize, no flow":
// interpolate_frames.
class FrameInterpolator {
public:
    virtual ~FrameInterpolator() = default;
    virtual int initialize(uint32_t width, uint32_t height) = 0;
    virtual int interpolate(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, const float *flow,
                            uint32_t w, uint32_t h, float t, uint8_t *out, size_t out_cap) = 0;
    virtual int interpolate_frames(const uint8_t *frame1, size_t len1, const uint8_t *frame2, size_t len2, float t,
                                   uint8_t *out, size_t out_cap) = 0;
    virtual const char *name() const = 0;
    virtual int set_quality(InterpolationQuality q) = 0;
    virtual InterpolationQuality quality() const = 0;
    virtual const char *last_error() const = 0;
};

class HipFrameInterpolator final : public FrameInterpolator {
public:
    explicit HipFrameInterpolator(int wg_preset);
    ~HipFrameInterpolator() override;
    HipFrameInterpolator(const HipFrameInterpolator &) = delete;
    HipFrameInterpolator &operator=(const HipFrameInterpolator &) = delete;

    // allocates the device buffers and pinned staging for width x height frames up front (interpolation/mod.rs:305-452
    // builds its buffers here); calling it again with the same size is a no-op, as there
    int initialize(uint32_t width, uint32_t height) override;
    int interpolate(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, const float *flow, uint32_t w,
                    uint32_t h, float t, uint8_t *out, size_t out_cap) override;
    // the trait's own form: frames of the initialize() size, zero flow ("Interpolator not initialized" before it:
    // interpolation/mod.rs:368-370)
    int interpolate_frames(const uint8_t *frame1, size_t len1, const uint8_t *frame2, size_t len2, float t, uint8_t *out,
                           size_t out_cap) override;
    // the quality level is kept and reported; like the upscaler's (upscale/mod.rs:1072-1077) it does not change the arithmetic
    int set_quality(InterpolationQuality q) override;
    InterpolationQuality quality() const override { return quality_; }
    int interpolate_device(const void *d_a, size_t a_stride, const void *d_b, size_t b_stride, const void *d_flow,
                           uint32_t w, uint32_t h, float t, void *d_out, uint32_t n_pairs, hipStream_t stream);
    const char *name() const override { return "HipWarpBlendInterpolator"; }
    const char *last_error() const override { return error_.c_str(); }
    int set_device(int device);
    int set_input_format(int format); // nus_pixel_format of both frames; output RGBA8
    int set_flow_format(int format);  // nus_flow_format of the device flow field
    // nus_interp_mode of the dense-flow warp: 0 EXACT (the CPU's separate roundings, bit-exact against the oracle; default),
    // 1 FMA (fused lerps, +-1 LSB: the interpolation path's contract).  The zero-flow blend is exact in both.
    int set_mode(int mode);
    int mode() const { return fma_ ? 1 : 0; }
    bool last_gpu_ms(double *ms) const;
    int wg_preset() const { return wg_preset_; }

private:
    int fail(int status, const std::string &msg);
    int fail_hip(hipError_t e, const char *what);
    int ensure(size_t frame_bytes, bool with_flow);
    void release();

    mutable std::mutex mu_;
    int wg_preset_;
    InterpolationQuality quality_ = InterpolationQuality::Medium;
    uint32_t init_w_ = 0, init_h_ = 0; // initialize(): 0 = not called
    int device_ = 0;
    bool device_ready_ = false;
    int in_format_ = 0; // nus_pixel_format
    size_t cap_bytes_ = 0;
    bool cap_flow_ = false;
    uint8_t *d_a_ = nullptr, *d_b_ = nullptr, *d_out_ = nullptr;
    bool flow_half_ = false; // interpolate_device reads 2 x f16 per pixel
    bool fma_ = false;       // dense-flow warp in FMA mode
    float *d_flow_ = nullptr;
    uint8_t *h_stage_ = nullptr; // pinned: a | b | out
    float *h_flow_ = nullptr;    // pinned
    hipStream_t stream_ = nullptr;
    hipEvent_t k_begin_ = nullptr, k_end_ = nullptr, half_done_ = nullptr;
    bool have_ms_ = false;
    double last_ms_ = 0.0;
    std::string error_;
};

} // namespace nus

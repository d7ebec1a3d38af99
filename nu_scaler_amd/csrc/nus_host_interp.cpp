// nus_host_interp.cpp -- HipFrameInterpolator: the host side of the two-frame warp + blend path
// (WgpuFrameInterpolator::interpolate_py, nu_scaler_core/src/wgpu_interpolator.rs:215-491).  See nus_host.hpp.
#include "nus_host.hpp"

#include "nus_copy.hpp"
#include "nus_host_util.hpp"

#include <cstring>

namespace nus {

// ---------------------------------------------------------------------------------
// HipFrameInterpolator
// ---------------------------------------------------------------------------------

HipFrameInterpolator::HipFrameInterpolator(int wg_preset) : wg_preset_(wg_preset) {}

HipFrameInterpolator::~HipFrameInterpolator()
{
    release();
    if (device_ready_) {
        if (k_begin_) (void)hipEventDestroy(k_begin_);
        if (k_end_) (void)hipEventDestroy(k_end_);
        if (half_done_) (void)hipEventDestroy(half_done_);
        if (stream_) (void)hipStreamDestroy(stream_);
    }
}

int HipFrameInterpolator::fail(int status, const std::string &msg)
{
    error_ = msg;
    set_thread_error(msg);
    return status;
}

int HipFrameInterpolator::fail_hip(hipError_t e, const char *what)
{
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? kOutOfMemory : kHipError,
                fmt("HIP error in %s: %s", what, hipGetErrorString(e)));
}

int HipFrameInterpolator::set_device(int device)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (device < 0) return fail(kInvalidArgument, "negative device index");
    if (device_ready_) return fail(kInvalidArgument, "set_device must precede the first interpolation");
    device_ = device;
    return kOk;
}

void HipFrameInterpolator::release()
{
    if (!device_ready_) return;
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (d_a_) (void)hipFree(d_a_);
    if (d_b_) (void)hipFree(d_b_);
    if (d_out_) (void)hipFree(d_out_);
    if (d_flow_) (void)hipFree(d_flow_);
    pinned_free(h_stage_);
    pinned_free(h_flow_);
    d_a_ = d_b_ = d_out_ = nullptr;
    d_flow_ = nullptr;
    h_stage_ = nullptr;
    h_flow_ = nullptr;
    cap_bytes_ = 0;
    cap_flow_ = false;
}

int HipFrameInterpolator::ensure(size_t frame_bytes, bool with_flow)
{
    if (!device_ready_) {
        const int n = device_count();
        if (n <= 0) return fail(kNoDevice, "no HIP device available (the gfx950 path has no CPU fallback)");
        if (device_ >= n) return fail(kNoDevice, fmt("HIP device %d requested but only %d present", device_, n));
        NUS_HIP(hipSetDevice(device_));
        NUS_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        NUS_HIP(hipEventCreate(&k_begin_));
        NUS_HIP(hipEventCreate(&k_end_));
        NUS_HIP(hipEventCreateWithFlags(&half_done_, hipEventDisableTiming));
        device_ready_ = true;
    }
    NUS_HIP(hipSetDevice(device_));
    if (frame_bytes > cap_bytes_ || (with_flow && !cap_flow_)) {
        // the reference reallocates its textures on every call (wgpu_interpolator.rs:253-321);
        // here buffers persist and only grow.
        const size_t want = frame_bytes > cap_bytes_ ? frame_bytes : cap_bytes_;
        const bool flow = with_flow || cap_flow_;
        release();
        NUS_HIP(hipMalloc(reinterpret_cast<void **>(&d_a_), want));
        NUS_HIP(hipMalloc(reinterpret_cast<void **>(&d_b_), want));
        NUS_HIP(hipMalloc(reinterpret_cast<void **>(&d_out_), want));
        NUS_HIP(pinned_alloc(reinterpret_cast<void **>(&h_stage_), want * 3));
        if (flow) {
            NUS_HIP(hipMalloc(reinterpret_cast<void **>(&d_flow_), want * 2));
            NUS_HIP(pinned_alloc(reinterpret_cast<void **>(&h_flow_), want * 2));
        }
        cap_bytes_ = want;
        cap_flow_ = flow;
    }
    return kOk;
}

int HipFrameInterpolator::initialize(uint32_t width, uint32_t height)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 31))
        return fail(kInvalidArgument, "initialize: bad dimensions");
    if (init_w_ == width && init_h_ == height) return kOk; // interpolation/mod.rs:306-308
    const int rc = ensure((size_t)width * height * 4, false);
    if (rc != kOk) return rc;
    init_w_ = width;
    init_h_ = height;
    error_.clear();
    return kOk;
}

int HipFrameInterpolator::interpolate_frames(const uint8_t *frame1, size_t len1, const uint8_t *frame2, size_t len2, float t,
                                             uint8_t *out, size_t out_cap)
{
    uint32_t w, h;
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (init_w_ == 0) return fail(kNotInitialized, "Interpolator not initialized"); // interpolation/mod.rs:368-370
        w = init_w_;
        h = init_h_;
    }
    return interpolate(frame1, len1, frame2, len2, nullptr, w, h, t, out, out_cap);
}

int HipFrameInterpolator::set_quality(InterpolationQuality q)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (q != InterpolationQuality::High && q != InterpolationQuality::Medium && q != InterpolationQuality::Low)
        return fail(kInvalidArgument, "unknown interpolation quality");
    quality_ = q;
    return kOk;
}

int HipFrameInterpolator::interpolate(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, const float *flow,
                                      uint32_t w, uint32_t h, float t, uint8_t *out, size_t out_cap)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (w == 0 || h == 0 || (uint64_t)w * h >= (1ull << 31)) return fail(kInvalidArgument, "interpolate: bad dimensions");
    const size_t expected = (size_t)w * h * 4;
    if (a_len != expected || b_len != expected)
        // wgpu_interpolator.rs:234-237
        return fail(kSizeMismatch, fmt("Expected %zu bytes per frame for %ux%ux4 RGBA, got frame_a: %zu bytes, frame_b: %zu bytes",
                                       expected, w, h, a_len, b_len));
    if (!a || !b || !out) return fail(kInvalidArgument, "interpolate: null frame pointer");
    if (out_cap < expected) return fail(kInvalidArgument, "interpolate: output capacity too small");
    int rc = ensure(expected, flow != nullptr);
    if (rc != kOk) return rc;
    uint8_t *ha = h_stage_, *hb = h_stage_ + cap_bytes_, *ho = h_stage_ + 2 * cap_bytes_;
    // while the pair is staged, uploaded and on the GPU, idle workers of the copy pool make the pages of the result buffer
    // present (a buffer fresh from the allocator -- what interpolate_py returns -- otherwise takes its faults in the copy-out)
    struct Populate {
        CopyTicket t;
        ~Populate() { parallel_copy_wait(t); } // on every way out: queued requests point into `out`
    } populate;
    if (parallel_populate_prepare(out, expected)) parallel_populate_async(out, expected, populate.t);
    // stage A, start its DMA, stage B meanwhile (the reference uploads both synchronously:
    // wgpu_interpolator.rs:253-321)
    parallel_copy(ha, a, expected);
    NUS_HIP(hipMemcpyAsync(d_a_, ha, expected, hipMemcpyHostToDevice, stream_));
    parallel_copy(hb, b, expected);
    NUS_HIP(hipMemcpyAsync(d_b_, hb, expected, hipMemcpyHostToDevice, stream_));
    if (flow) {
        parallel_copy(h_flow_, flow, expected * 2);
        NUS_HIP(hipMemcpyAsync(d_flow_, h_flow_, expected * 2, hipMemcpyHostToDevice, stream_));
    }
    WarpLaunch L;
    L.a = d_a_;
    L.b = d_b_;
    L.flow = flow ? d_flow_ : nullptr;
    L.out = d_out_;
    L.a_stride = L.b_stride = expected;
    L.w = w;
    L.h = h;
    L.t = t;
    L.n_pairs = 1;
    L.stream = stream_;
    L.fma = fma_;
    L.in_sel = input_selector(in_format_);
    NUS_HIP(hipEventRecord(k_begin_, stream_));
    hipError_t e = launch_warp_blend(L);
    if (e != hipSuccess) return fail_hip(e, "warp+blend launch");
    NUS_HIP(hipEventRecord(k_end_, stream_));
    // the frame comes back in two halves: the host copy of the first overlaps the DMA of the second
    const size_t half = (expected / 2 + 4095) & ~(size_t)4095;
    const size_t first = half < expected ? half : expected;
    NUS_HIP(hipMemcpyAsync(ho, d_out_, first, hipMemcpyDeviceToHost, stream_));
    NUS_HIP(hipEventRecord(half_done_, stream_));
    if (first < expected) NUS_HIP(hipMemcpyAsync(ho + first, d_out_ + first, expected - first, hipMemcpyDeviceToHost, stream_));
    NUS_HIP(hipEventSynchronize(half_done_));
    parallel_copy(out, ho, first);
    NUS_HIP(hipStreamSynchronize(stream_));
    if (first < expected) parallel_copy(out + first, ho + first, expected - first);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, k_begin_, k_end_) == hipSuccess) {
        have_ms_ = true;
        last_ms_ = ms;
    } else {
        (void)hipGetLastError();
    }
    return kOk;
}

int HipFrameInterpolator::interpolate_device(const void *d_a, size_t a_stride, const void *d_b, size_t b_stride,
                                             const void *d_flow, uint32_t w, uint32_t h, float t, void *d_out,
                                             uint32_t n_pairs, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (w == 0 || h == 0 || (uint64_t)w * h >= (1ull << 31)) return fail(kInvalidArgument, "interpolate_device: bad dimensions");
    if (!d_a || !d_b || !d_out) return fail(kInvalidArgument, "interpolate_device: null device pointer");
    if (n_pairs == 0) return kOk;
    if ((reinterpret_cast<uintptr_t>(d_a) % 4) || (reinterpret_cast<uintptr_t>(d_b) % 4) ||
        (reinterpret_cast<uintptr_t>(d_out) % 4) || (a_stride % 4) || (b_stride % 4) ||
        (d_flow && reinterpret_cast<uintptr_t>(d_flow) % (flow_half_ ? 4 : 8)))
        return fail(kInvalidArgument, "interpolate_device: pointers/strides must be pixel aligned");
    const int n = device_count();
    if (n <= 0) return fail(kNoDevice, "no HIP device available (the gfx950 path has no CPU fallback)");
    NUS_HIP(hipSetDevice(device_));
    WarpLaunch L;
    L.a = static_cast<const uint8_t *>(d_a);
    L.b = static_cast<const uint8_t *>(d_b);
    L.flow = static_cast<const float *>(d_flow);
    L.flow_half = flow_half_;
    L.fma = fma_;
    L.out = static_cast<uint8_t *>(d_out);
    L.a_stride = a_stride;
    L.b_stride = b_stride;
    L.w = w;
    L.h = h;
    L.t = t;
    L.n_pairs = n_pairs;
    L.stream = stream;
    L.in_sel = input_selector(in_format_);
    hipError_t e = launch_warp_blend(L);
    if (e != hipSuccess) return fail_hip(e, "warp+blend launch");
    return kOk;
}

int HipFrameInterpolator::set_mode(int mode)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (mode != 0 && mode != 1) return fail(kInvalidArgument, "unknown interpolation mode");
    fma_ = mode == 1;
    return kOk;
}

int HipFrameInterpolator::set_flow_format(int format)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (format != 0 && format != 1) return fail(kInvalidArgument, "unknown flow format");
    flow_half_ = format == 1;
    return kOk;
}

int HipFrameInterpolator::set_input_format(int format)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (format < 0 || format > 3) return fail(kInvalidArgument, "unknown input format");
    in_format_ = format;
    return kOk;
}

bool HipFrameInterpolator::last_gpu_ms(double *ms) const
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!have_ms_) return false;
    if (ms) *ms = last_ms_;
    return true;
}

} // namespace nus

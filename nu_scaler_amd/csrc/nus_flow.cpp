// nus_flow.cpp -- see nus_flow.hpp.
#include "nus_flow.hpp"

#include <cstdlib>

#include <cstring>

#include "nus_host.hpp"
#include "nus_kernels.hpp"
#include "nus_transfer.hpp"

namespace nus {

namespace {
int device_count_()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
} // namespace

#define NUS_HIP(call)                                     \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return fail_hip(e_, #call); \
    } while (0)

// The host entry points move their images through the library's own pinned ring (nus_transfer.hpp): a caller's pageable
// buffer is never handed to the runtime's copy.
#define NUS_XFER(call)                                        \
    do {                                                      \
        const int x_ = (call);                                \
        if (x_ != kOk) return fail(x_, thread_error());       \
    } while (0)

HipFlowEstimator::~HipFlowEstimator() { release(); }

int HipFlowEstimator::fail(int status, const std::string &msg)
{
    error_ = msg;
    set_thread_error(msg);
    return status;
}

int HipFlowEstimator::fail_hip(hipError_t e, const char *what)
{
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? kOutOfMemory : kHipError,
                std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

int HipFlowEstimator::set_tiled(int mode)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (mode < 0 || mode > 3) return fail(kInvalidArgument, "set_tiled: 0 plain, 1 multi-step (kernel chosen by size), 2 LDS tiles, 3 streamed");
    tiled_ = mode != 0;
    jacobi_ = mode == 2 ? kJacobiTiles : (mode == 3 ? kJacobiStream : kJacobiAuto);
    return kOk;
}

int HipFlowEstimator::set_mode(int mode)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (mode != 0 && mode != 1) return fail(kInvalidArgument, "set_mode: 0 exact, 1 fast");
    fast_ = mode == 1;
    return kOk;
}

int HipFlowEstimator::set_device(int device)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (device < 0) return fail(kInvalidArgument, "negative device index");
    if (ready_) return fail(kInvalidArgument, "set_device must precede the first call");
    device_ = device;
    return kOk;
}

int HipFlowEstimator::ensure_device()
{
    const int n = device_count_();
    if (n <= 0) return fail(kNoDevice, "no HIP device available (the gfx950 path has no CPU fallback)");
    if (device_ >= n) return fail(kNoDevice, "requested HIP device not present");
    NUS_HIP(hipSetDevice(device_));
    if (!ready_) {
        NUS_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        ready_ = true;
    }
    return kOk;
}

int HipFlowEstimator::reserve(size_t bytes, int slot)
{
    if (bytes <= slot_cap_[slot]) return kOk;
    if (slot_[slot]) {
        NUS_HIP(hipStreamSynchronize(stream_));
        NUS_HIP(hipFree(slot_[slot]));
        slot_[slot] = nullptr;
        slot_cap_[slot] = 0;
    }
    NUS_HIP(hipMalloc(&slot_[slot], bytes));
    slot_cap_[slot] = bytes;
    return kOk;
}

void HipFlowEstimator::release()
{
    if (!ready_) return;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    for (int i = 0; i < kSlotCount; ++i) {
        if (slot_[i]) (void)hipFree(slot_[i]);
        slot_[i] = nullptr;
        slot_cap_[i] = 0;
    }
    (void)hipStreamDestroy(stream_);
    ready_ = false;
}

#define CHECK_DIMS(w, h)                                                             \
    if ((w) == 0 || (h) == 0 || (uint64_t)(w) * (h) >= (1ull << 28))                 \
        return fail(kInvalidArgument, "flow: bad image dimensions");

int HipFlowEstimator::rgba8_to_f32(const uint8_t *in, uint32_t w, uint32_t h, float *out)
{
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_DIMS(w, h);
    if (!in || !out) return fail(kInvalidArgument, "flow: null pointer");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    const size_t npx = (size_t)w * h;
    if ((rc = reserve(npx * 4, 0)) != kOk || (rc = reserve(npx * 16, 1)) != kOk) return rc;
    NUS_XFER(upload(slot_[0], in, npx * 4, stream_));
    NUS_HIP(launch_rgba8_to_f32(static_cast<const uint8_t *>(slot_[0]), static_cast<float *>(slot_[1]), w, h, stream_));
    NUS_XFER(download(out, slot_[1], npx * 16, stream_));
    NUS_HIP(hipStreamSynchronize(stream_));
    return kOk;
}

int HipFlowEstimator::blur(const float *in, uint32_t w, uint32_t h, float *out)
{
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_DIMS(w, h);
    if (!in || !out) return fail(kInvalidArgument, "flow: null pointer");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    const size_t bytes = (size_t)w * h * 16;
    if ((rc = reserve(bytes, 0)) != kOk || (rc = reserve(bytes, 1)) != kOk) return rc;
    float *d0 = static_cast<float *>(slot_[0]), *d1 = static_cast<float *>(slot_[1]);
    NUS_XFER(upload(d0, in, bytes, stream_));
    NUS_HIP(launch_blur(d0, d1, w, h, true, stream_));
    NUS_HIP(launch_blur(d1, d0, w, h, false, stream_));
    NUS_XFER(download(out, d0, bytes, stream_));
    NUS_HIP(hipStreamSynchronize(stream_));
    return kOk;
}

int HipFlowEstimator::downsample(const float *in, uint32_t w, uint32_t h, float *out)
{
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_DIMS(w, h);
    if (!in || !out) return fail(kInvalidArgument, "flow: null pointer");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    const size_t bytes = (size_t)w * h * 16, obytes = (size_t)((w + 1) / 2) * ((h + 1) / 2) * 16;
    if ((rc = reserve(bytes, 0)) != kOk || (rc = reserve(obytes, 1)) != kOk) return rc;
    NUS_XFER(upload(slot_[0], in, bytes, stream_));
    NUS_HIP(launch_downsample(static_cast<const float *>(slot_[0]), static_cast<float *>(slot_[1]), w, h, stream_));
    NUS_XFER(download(out, slot_[1], obytes, stream_));
    NUS_HIP(hipStreamSynchronize(stream_));
    return kOk;
}

int HipFlowEstimator::horn_schunck(const float *i1, const float *i2, const float *flow_in, uint32_t w, uint32_t h,
                                   float lambda, uint32_t iterations, float *flow_out)
{
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_DIMS(w, h);
    if (!i1 || !i2 || !flow_out) return fail(kInvalidArgument, "flow: null pointer");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    const size_t ib = (size_t)w * h * 16, fb = (size_t)w * h * 8;
    if ((rc = reserve(ib, 0)) != kOk || (rc = reserve(ib, 1)) != kOk || (rc = reserve(fb, 2)) != kOk ||
        (rc = reserve(fb, 3)) != kOk)
        return rc;
    NUS_XFER(upload(slot_[0], i1, ib, stream_));
    NUS_XFER(upload(slot_[1], i2, ib, stream_));
    if (flow_in)
        NUS_XFER(upload(slot_[2], flow_in, fb, stream_));
    else
        NUS_HIP(hipMemsetAsync(slot_[2], 0, fb, stream_)); // compute_coarse_flow clears the flow (:1136-1154)
    float *f0 = static_cast<float *>(slot_[2]), *f1 = static_cast<float *>(slot_[3]);
    if (tiled_) {
        if ((rc = reserve(ib, 4)) != kOk) return rc; // 3 floats of coefficients per cell
        float *coef = static_cast<float *>(slot_[4]);
        NUS_HIP(launch_hs_prepare(static_cast<const float *>(slot_[0]), static_cast<const float *>(slot_[1]), false, coef, w, h, stream_));
        NUS_HIP(launch_hs_iterate(coef, lambda, &f0, &f1, w, h, iterations, false, nullptr, stream_, 1, 0, 0, 0, jacobi_));
    } else {
        for (uint32_t i = 0; i < iterations; ++i) { // ping-pong as :1156-1193
            NUS_HIP(launch_horn_schunck(static_cast<const float *>(slot_[0]), static_cast<const float *>(slot_[1]), f0, f1, w, h, lambda, stream_));
            float *t = f0;
            f0 = f1;
            f1 = t;
        }
    }
    NUS_XFER(download(flow_out, f0, fb, stream_));
    NUS_HIP(hipStreamSynchronize(stream_));
    return kOk;
}

int HipFlowEstimator::upsample(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh, float scale)
{
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_DIMS(sw, sh);
    CHECK_DIMS(dw, dh);
    if (!src || !dst) return fail(kInvalidArgument, "flow: null pointer");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    const size_t sb = (size_t)sw * sh * 8, db = (size_t)dw * dh * 8;
    if ((rc = reserve(sb, 2)) != kOk || (rc = reserve(db, 3)) != kOk) return rc;
    NUS_XFER(upload(slot_[2], src, sb, stream_));
    NUS_HIP(launch_flow_upsample(static_cast<const float *>(slot_[2]), sw, sh, static_cast<float *>(slot_[3]), dw, dh, scale, stream_));
    NUS_XFER(download(dst, slot_[3], db, stream_));
    NUS_HIP(hipStreamSynchronize(stream_));
    return kOk;
}

// Device workspace layout of one estimate (slots): 0 tmp / current input (w*h*16),
// 1 blur temp (w*h*16), 2..3 flow ping-pong (w*h*8), 4 pyramid A, 5 pyramid B (all levels,
// packed), 6..7 RGBA8 staging for the host entry point.
int HipFlowEstimator::plan(uint32_t w, uint32_t h, uint32_t levels, Pyramid &g)
{
    if (levels == 0 || levels > 12) return fail(kInvalidArgument, "flow: levels must be 1..12");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    // level geometry (build_pyramid: next = (cur + 1) / 2, wgpu_interpolator.rs:1008-1009)
    g.total = 0;
    g.levels = 0;
    for (uint32_t l = 0, cw = w, ch = h; l < levels; ++l) {
        g.w[l] = cw;
        g.h[l] = ch;
        g.offset[l] = g.total;
        g.total += (size_t)cw * ch * 16;
        g.levels = l + 1;
        if (cw == 1 && ch == 1) break;
        cw = (cw + 1) / 2;
        ch = (ch + 1) / 2;
    }
    const size_t ib = (size_t)w * h * 16, fb = (size_t)w * h * 8;
    // slot 1 doubles as the per-level coefficient buffer (3 floats per cell) once the pyramids exist
    if ((rc = reserve(ib, 0)) != kOk || (rc = reserve(ib, 1)) != kOk || (rc = reserve(fb, 2)) != kOk ||
        (rc = reserve(fb, 3)) != kOk || (rc = reserve(g.total, 4)) != kOk || (rc = reserve(g.total, 5)) != kOk)
        return rc;
    return kOk;
}

// Pyramid of one RGBA8 frame into slot `pyr_slot` (4 or 5).
int HipFlowEstimator::build_pyramid(const void *frame, int pyr_slot, const Pyramid &g, hipStream_t stream)
{
    float *cur = static_cast<float *>(slot_[0]), *tmp = static_cast<float *>(slot_[1]);
    uint8_t *pyr = static_cast<uint8_t *>(slot_[pyr_slot]);
    if (tiled_) {
        // fused level kernel: writes the level's luminance plane (at the level's offset; the
        // f32 RGBA level itself is not needed) and the downsampled input of level l+1, which
        // ping-pongs between cur and tmp
        float *nxt[2] = {cur, tmp};
        const void *src = frame;
        for (uint32_t l = 0; l < g.levels; ++l) {
            float *level = reinterpret_cast<float *>(pyr + g.offset[l]);
            float *next = l + 1 < g.levels ? nxt[l & 1] : nullptr;
            NUS_HIP(launch_pyramid_level(src, l == 0, level, next, g.w[l], g.h[l], stream, 1, 0, 0, 0, jacobi_));
            src = next;
        }
        return kOk;
    }
    NUS_HIP(launch_rgba8_to_f32(static_cast<const uint8_t *>(frame), cur, g.w[0], g.h[0], stream));
    for (uint32_t l = 0; l < g.levels; ++l) {
        float *level = reinterpret_cast<float *>(pyr + g.offset[l]);
        NUS_HIP(launch_blur(cur, tmp, g.w[l], g.h[l], true, stream));
        NUS_HIP(launch_blur(tmp, level, g.w[l], g.h[l], false, stream));
        if (l + 1 < g.levels) NUS_HIP(launch_downsample(level, cur, g.w[l], g.h[l], stream));
    }
    return kOk;
}

// Coarse-to-fine Horn-Schunck between the pyramids in slots `slot_a` and `slot_b`.
int HipFlowEstimator::solve(int slot_a, int slot_b, const Pyramid &g, uint32_t coarse_iters, uint32_t refine_iters,
                            float lambda, void *d_flow_out, hipStream_t stream)
{
    int rc;
    const uint8_t *pa = static_cast<const uint8_t *>(slot_[slot_a]), *pb = static_cast<const uint8_t *>(slot_[slot_b]);
    float *f0 = static_cast<float *>(slot_[2]), *f1 = static_cast<float *>(slot_[3]);
    float *coef = static_cast<float *>(slot_[1]); // the blur temp is free once the pyramids exist
    const uint32_t L = g.levels - 1;
    // compute_coarse_flow starts from zero flow (:1136-1154): the tiled kernel takes that as a null input;
    // the last launch of the finest level writes the caller's buffer directly
    bool zero = true;
    auto iterate = [&](uint32_t l, uint32_t iters, bool prepared) -> int {
        const float *i1 = reinterpret_cast<const float *>(pa + g.offset[l]), *i2 = reinterpret_cast<const float *>(pb + g.offset[l]);
        if (iters == 0) return kOk;
        if (tiled_) {
            if (!prepared) NUS_HIP(launch_hs_prepare(i1, i2, true, coef, g.w[l], g.h[l], stream)); // tiled pyramids hold luminance planes
            NUS_HIP(launch_hs_iterate(coef, lambda, &f0, &f1, g.w[l], g.h[l], iters, zero,
                                      l == 0 ? static_cast<float *>(d_flow_out) : nullptr, stream, 1, 0, 0, 0, jacobi_));
            zero = false;
            return kOk;
        }
        for (uint32_t i = 0; i < iters; ++i) {
            NUS_HIP(launch_horn_schunck(i1, i2, f0, f1, g.w[l], g.h[l], lambda, stream));
            float *t = f0;
            f0 = f1;
            f1 = t;
        }
        return kOk;
    };
    if (!tiled_ || coarse_iters == 0) {
        NUS_HIP(hipMemsetAsync(f0, 0, (size_t)g.w[L] * g.h[L] * 8, stream));
        zero = false;
    }
    if ((rc = iterate(L, coarse_iters, false)) != kOk) return rc;
    for (int l = (int)L - 1; l >= 0; --l) {
        const bool fused_setup = tiled_ && refine_iters > 0; // the level's derivatives and the upsampled flow in one launch
        if (fused_setup)
            NUS_HIP(launch_hs_level_setup(reinterpret_cast<const float *>(pa + g.offset[l]), reinterpret_cast<const float *>(pb + g.offset[l]),
                                          coef, g.w[l], g.h[l], f0, g.w[l + 1], g.h[l + 1], f1, 2.0f, stream));
        else
            NUS_HIP(launch_flow_upsample(f0, g.w[l + 1], g.h[l + 1], f1, g.w[l], g.h[l], 2.0f, stream));
        float *t = f0;
        f0 = f1;
        f1 = t;
        if ((rc = iterate((uint32_t)l, refine_iters, fused_setup)) != kOk) return rc;
    }
    if (f0 != d_flow_out)
        NUS_HIP(hipMemcpyAsync(d_flow_out, f0, (size_t)g.w[0] * g.h[0] * 8, hipMemcpyDeviceToDevice, stream));
    return kOk;
}

int HipFlowEstimator::estimate_device(const void *d_a, const void *d_b, uint32_t w, uint32_t h, uint32_t levels,
                                      uint32_t coarse_iters, uint32_t refine_iters, float lambda, void *d_flow_out,
                                      hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_DIMS(w, h);
    if (!d_a || !d_b || !d_flow_out) return fail(kInvalidArgument, "flow: null device pointer");
    Pyramid g;
    int rc = plan(w, h, levels, g);
    if (rc != kOk) return rc;
    // FAST arithmetic lives in the batch solver's register-pipelined kernels, which need a batch (or a frame) big enough to fill
    // the GPU; a single 1080p pair is 576 waves -- the LDS-tile kernels of the exact path are 2.7x faster there (220 against
    // 605 us), and their result satisfies FAST's contract trivially.  So a pair alone takes the FAST kernels only where the
    // finest level would stream anyway, or where the streamed kernel is forced (set_tiled(3)).
    if (fast_ && (jacobi_ == kJacobiStream || (jacobi_ == kJacobiAuto && hs_iterate_streams(g.w[0], g.h[0], 1, kJacobiAuto)))) {
        const size_t fb = (size_t)w * h * 4;
        if ((rc = reserve(2 * fb, 8)) != kOk) return rc;
        uint8_t *two = static_cast<uint8_t *>(slot_[8]);
        NUS_HIP(hipMemcpyAsync(two, d_a, fb, hipMemcpyDeviceToDevice, stream));
        NUS_HIP(hipMemcpyAsync(two + fb, d_b, fb, hipMemcpyDeviceToDevice, stream));
        return solve_batch(two, 1, g, coarse_iters, refine_iters, lambda, static_cast<uint8_t *>(d_flow_out), stream);
    }
    if ((rc = build_pyramid(d_a, 4, g, stream)) != kOk || (rc = build_pyramid(d_b, 5, g, stream)) != kOk) return rc;
    return solve(4, 5, g, coarse_iters, refine_iters, lambda, d_flow_out, stream);
}

// Flows between consecutive frames of a device-resident stream: frame k+1's pyramid, built for the
// pair (k, k+1), is frame A's pyramid of the pair (k+1, k+2), so each frame's pyramid is built once.
int HipFlowEstimator::estimate_device_stream(const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t h,
                                             uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda,
                                             void *d_flows, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!d_flows) return fail(kInvalidArgument, "flow: null device pointer");
    return stream_impl(d_frames, n_frames, w, h, levels, coarse_iters, refine_iters, lambda, d_flows, nullptr, 0.5f, stream);
}

// flow_half: the flows between estimator and warp -- and at d_flows, if given -- as 2 x IEEE half per pixel (Rg16Float, the
// reference's live flow layout: wgpu_interpolator.rs:276), each the f32 flow rounded to nearest even; the warp reads them as such.
int HipFlowEstimator::interpolate_device_stream(const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels,
                                                uint32_t coarse_iters, uint32_t refine_iters, float lambda, float t, void *d_flows,
                                                void *d_mid, hipStream_t stream, bool flow_half)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!d_mid) return fail(kInvalidArgument, "flow: null device pointer");
    if (!(t >= 0.0f && t <= 1.0f)) return fail(kInvalidArgument, "flow: t must be in [0, 1]");
    if ((reinterpret_cast<uintptr_t>(d_mid) % 16) || (reinterpret_cast<uintptr_t>(d_flows) % 16))
        return fail(kInvalidArgument, "flow: device pointers must be 16-byte aligned");
    return stream_impl(d_frames, n_frames, w, h, levels, coarse_iters, refine_iters, lambda, d_flows, d_mid, t, stream, flow_half);
}

// (called with mu_ held)  d_flows may be null when d_mid is not: the caller wants the in-between frames only.
int HipFlowEstimator::stream_impl(const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels,
                                  uint32_t coarse_iters, uint32_t refine_iters, float lambda, void *d_flows, void *d_mid, float t,
                                  hipStream_t stream, bool flow_half)
{
    CHECK_DIMS(w, h);
    if (!d_frames || (!d_flows && !d_mid)) return fail(kInvalidArgument, "flow: null device pointer");
    if (n_frames < 2) return fail(kInvalidArgument, "flow: a stream needs at least 2 frames");
    Pyramid g;
    int rc = plan(w, h, levels, g);
    if (rc != kOk) return rc;
    const uint8_t *frames = static_cast<const uint8_t *>(d_frames);
    uint8_t *flows = static_cast<uint8_t *>(d_flows), *mid = static_cast<uint8_t *>(d_mid);
    const size_t frame_bytes = (size_t)w * h * 4, flow_bytes = (size_t)w * h * (flow_half ? 4 : 8);
    // the warp kernel behind an estimator that did not warp itself: pairs [k0, k0 + n) with the flows at `fl`
    auto warp_behind = [&](uint32_t k0, uint32_t n, const void *fl) -> int {
        WarpLaunch L;
        L.a = frames + (size_t)k0 * frame_bytes;
        L.b = L.a + frame_bytes;
        L.a_stride = L.b_stride = frame_bytes;
        L.flow = static_cast<const float *>(fl);
        L.flow_half = flow_half;
        L.fma = true;
        L.out = mid + (size_t)k0 * frame_bytes;
        L.w = w, L.h = h, L.t = t, L.n_pairs = n, L.stream = stream;
        NUS_HIP(launch_warp_blend(L));
        return kOk;
    };
    if (!tiled_ && !fast_) { // the shader-shaped kernels, pair by pair (each frame's pyramid still built once)
        const size_t f32_bytes = (size_t)w * h * 8;
        if ((mid && !flows) || flow_half) { // (they write a pair's f32 flow where they are told to: one pair's worth of workspace)
            if ((rc = reserve(f32_bytes, 9)) != kOk) return rc;
        }
        if (flow_half && !flows && (rc = reserve(flow_bytes, 10)) != kOk) return rc;
        if ((rc = build_pyramid(frames, 4, g, stream)) != kOk) return rc;
        for (uint32_t k = 0; k + 1 < n_frames; ++k) {
            const int slot_a = 4 + (int)(k & 1), slot_b = 5 - (int)(k & 1);
            if ((rc = build_pyramid(frames + (size_t)(k + 1) * frame_bytes, slot_b, g, stream)) != kOk) return rc;
            void *fl = flows && !flow_half ? static_cast<void *>(flows + (size_t)k * flow_bytes) : slot_[9];
            if ((rc = solve(slot_a, slot_b, g, coarse_iters, refine_iters, lambda, fl, stream)) != kOk) return rc;
            if (flow_half) {
                void *hf = flows ? static_cast<void *>(flows + (size_t)k * flow_bytes) : slot_[10];
                NUS_HIP(launch_flow_to_half(static_cast<const float *>(fl), hf, (size_t)w * h, stream));
                fl = hf;
            }
            if (mid && (rc = warp_behind(k, 1, fl)) != kOk) return rc;
        }
        return kOk;
    }
    // Tiled kernels: the pairs of a chunk go through every stage TOGETHER, one launch per stage with the pairs on the
    // grid's z axis -- a 480x270 level of one pair is 510 tiles (two per CU, latency bound); of 64 pairs it fills the GPU.
    const uint32_t n_pairs = n_frames - 1;
    // per pair: luminance planes 4/3 x 4 B, level inputs (1/4 + 1/16) x 16 B, two flows 16 B per pixel -- and 12 B of
    // coefficients if some level's Jacobi steps run on LDS tiles (the streamed kernel takes them from the planes)
    // (dev overrides, round 6 -- measured and left at the defaults: profiles/r06_flow_chunk_size.txt)
    const char *env_pairs = getenv("NUS_FLOW_MAX_CHUNK_PAIRS"), *env_gb = getenv("NUS_FLOW_WORKSPACE_GB");
    const uint32_t max_pairs = env_pairs && atoi(env_pairs) > 0 ? (uint32_t)atoi(env_pairs) : kStreamMaxChunkPairs;
    const size_t workspace = env_gb && atoi(env_gb) > 0 ? (size_t)atoi(env_gb) << 30 : kStreamWorkspaceBytes;
    auto chunk_for = [&](size_t bytes_per_pixel) {
        const uint32_t c = (uint32_t)(workspace / ((size_t)w * h * bytes_per_pixel));
        return c < 1 ? 1u : (c > max_pairs ? max_pairs : c);
    };
    uint32_t chunk = chunk_for(31);
    if (chunk > n_pairs) chunk = n_pairs;
    for (uint32_t l = 0; l < g.levels; ++l)
        if (!hs_iterate_streams(g.w[l], g.h[l], chunk, fast_ && jacobi_ == kJacobiStream ? kJacobiStreamFast : jacobi_)) {
            chunk = chunk_for(43);
            break;
        }
    for (uint32_t c0 = 0; c0 < n_pairs; c0 += chunk) {
        const uint32_t pairs = n_pairs - c0 < chunk ? n_pairs - c0 : chunk;
        if ((rc = solve_batch(frames + (size_t)c0 * frame_bytes, pairs, g, coarse_iters, refine_iters, lambda,
                              flows ? flows + (size_t)c0 * flow_bytes : nullptr, stream, mid ? mid + (size_t)c0 * frame_bytes : nullptr,
                              t, flow_half)) != kOk)
            return rc;
    }
    return kOk;
}

// `pairs` + 1 consecutive RGBA8 frames -> `pairs` flows, every stage one launch over the whole chunk.
// Workspace (grow-only slots): 0 / 1 the f32 RGBA inputs of the odd / even pyramid levels of all frames,
// 2 / 3 flow ping-pong [pair][level cells], 4 luminance planes [level][frame][cells], 5 coefficients [pair][cells][3].
// d_mid != nullptr: also the pairs' in-between frames at time t (see interpolate_device_stream); d_flows may then be null.
int HipFlowEstimator::solve_batch(const uint8_t *d_frames, uint32_t pairs, const Pyramid &g, uint32_t coarse_iters,
                                  uint32_t refine_iters, float lambda, uint8_t *d_flows, hipStream_t stream, uint8_t *d_mid, float t,
                                  bool flow_half)
{
    int rc;
    // The Jacobi kernel of a level.  FAST: k_hs_stream_fast where the level's batch would stream anyway (or the streamed kernel
    // is forced), the exact LDS-tile kernel -- on the FAST pyramid's planes -- where it would not (a small batch's coarse levels).
    auto level_kernel = [&](uint32_t l) -> int {
        if (!fast_ || jacobi_ == kJacobiTiles) return jacobi_;
        if (jacobi_ == kJacobiStream) return kJacobiStreamFast;
        return hs_iterate_streams(g.w[l], g.h[l], pairs, kJacobiAuto) ? kJacobiStreamFast : kJacobiTiles;
    };
    // The luminance-only pyramid streams rows per wave as well: where level 0 of the batch is too small for that, the exact
    // LDS-tile pyramid runs instead (its planes serve either Jacobi kernel).
    const bool fast_pyramid = fast_ && !getenv("NUS_FLOW_FAST_EXACT_PYRAMID") &&
                              (jacobi_ == kJacobiStream || level_kernel(0) == kJacobiStreamFast);
    const int jacobi = fast_ ? (fast_pyramid ? kJacobiStream : kJacobiTiles) : jacobi_; // (for the exact pyramid launcher's choice)
    const uint32_t nf = pairs + 1, nl = g.levels, L = nl - 1;
    size_t cells[12], lum_off[12], lum_total = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        cells[l] = (size_t)g.w[l] * g.h[l];
        lum_off[l] = lum_total;
        lum_total += cells[l] * nf;
    }
    const size_t in_odd = nl > 1 ? cells[1] : 0, in_even = nl > 2 ? cells[2] : 0; // largest level input each buffer holds
    if ((rc = reserve(in_odd * nf * 16, 0)) != kOk || (rc = reserve(in_even * nf * 16, 1)) != kOk ||
        (rc = reserve(cells[0] * pairs * 8, 2)) != kOk || (rc = reserve(cells[0] * pairs * 8, 3)) != kOk ||
        (rc = reserve(lum_total * 4, 4)) != kOk)
        return rc;
    // A level whose Jacobi steps run in the streamed kernel needs no coefficient planes: that kernel takes the
    // derivatives from the luminance planes of the pair's two frames (consecutive planes of the level) as it goes.
    auto from_planes = [&](uint32_t l) { return hs_iterate_streams(g.w[l], g.h[l], pairs, level_kernel(l)); };
    size_t coef_cells = 0;
    for (uint32_t l = 0; l < nl; ++l)
        if (!from_planes(l) && cells[l] > coef_cells) coef_cells = cells[l];
    if (coef_cells != 0 && (rc = reserve(coef_cells * pairs * 12, 5)) != kOk) return rc;
    float *level_in[2] = {static_cast<float *>(slot_[0]), static_cast<float *>(slot_[1])}; // input of level l: [(l - 1) & 1]
    float *lum = static_cast<float *>(slot_[4]), *coef = static_cast<float *>(slot_[5]);
    float *f0 = static_cast<float *>(slot_[2]), *f1 = static_cast<float *>(slot_[3]);
    // pyramids of all frames, level by level
    for (uint32_t l = 0; l < nl; ++l) {
        const void *src = l == 0 ? static_cast<const void *>(d_frames) : level_in[(l - 1) & 1];
        const size_t src_stride = l == 0 ? cells[0] * 4 /* bytes */ : cells[l] /* float4 */;
        float *next = l + 1 < nl ? level_in[l & 1] : nullptr;
        if (fast_pyramid) { // luminance only (NUS_FLOW_FAST_EXACT_PYRAMID: dev switch, bisecting): one float per pixel between the levels (the buffers are sized for four)
            NUS_HIP(launch_pyramid_level_fast(src, l == 0, lum + lum_off[l], next, g.w[l], g.h[l], stream, nf, src_stride, cells[l],
                                              l + 1 < nl ? cells[l + 1] : 0));
            continue;
        }
        NUS_HIP(launch_pyramid_level(src, l == 0, lum + lum_off[l], next, g.w[l], g.h[l], stream, nf, src_stride, cells[l],
                                     l + 1 < nl ? cells[l + 1] : 0, jacobi));
    }
    float *const out = reinterpret_cast<float *>(d_flows);
    // the finest level's last launch warps the pairs itself where it can (HsWarp); `warped` says whether it did
    HsWarp hw;
    hw.frames = d_frames, hw.frame_stride = cells[0] * 4, hw.mid = d_mid, hw.t = t, hw.sel = kSelRGBA;
    hw.out_half = flow_half ? 1u : 0u;
    bool warped = false, wrote_half = false;
    // Rg16Float hand-off: only the FAST streamed kernel's last launch stores halves itself.  The caller's buffer (4 bytes per cell
    // then) may be handed to the solver only if that launch is what finishes level 0 -- every other kernel writes 2 x f32 per cell
    // and must be kept in the workspace, its flow converted afterwards.
    const bool fast_last = level_kernel(0) == kJacobiStreamFast && (nl > 1 ? refine_iters > 0 : coarse_iters > 0);
    float *const solver_out = flow_half && !fast_last ? nullptr : out;
    // `coarse`: the level continues the flow of level l + 1 in f0, which the first launch upsamples as it loads it
    auto iterate = [&](uint32_t l, uint32_t iters, bool zero, bool coarse) -> int {
        bool did = false, did_half = false;
        NUS_HIP(launch_hs_iterate(coef, lambda, &f0, &f1, g.w[l], g.h[l], iters, zero, l == 0 ? solver_out : nullptr, stream, pairs,
                                  cells[l] * 3, cells[l], cells[0], level_kernel(l), from_planes(l) ? lum + lum_off[l] : nullptr, cells[l],
                                  coarse ? f0 : nullptr, coarse ? g.w[l + 1] : 0, coarse ? g.h[l + 1] : 0, 2.0f,
                                  coarse ? cells[l + 1] : 0, l == 0 && (d_mid || flow_half) ? &hw : nullptr, &did, &did_half));
        if (l == 0) warped = did, wrote_half = did_half;
        return kOk;
    };
    // coarsest level: from zero flow (compute_coarse_flow, :1136-1154)
    if (coarse_iters > 0) {
        if (!from_planes(L))
            NUS_HIP(launch_hs_prepare(lum + lum_off[L], lum + lum_off[L] + cells[L], true, coef, g.w[L], g.h[L], stream, pairs,
                                      cells[L], cells[L] * 3));
        if ((rc = iterate(L, coarse_iters, true, false)) != kOk) return rc;
    } else {
        NUS_HIP(hipMemsetAsync(f0, 0, cells[L] * pairs * 8, stream));
    }
    for (int l = (int)L - 1; l >= 0; --l) {
        const float *l1 = lum + lum_off[l];
        if (refine_iters > 0 && from_planes((uint32_t)l)) { // derivatives and upsampled flow both computed inside the Jacobi kernel
            if ((rc = iterate((uint32_t)l, refine_iters, false, true)) != kOk) return rc;
            continue;
        }
        if (refine_iters > 0)
            NUS_HIP(launch_hs_level_setup(l1, l1 + cells[l], coef, g.w[l], g.h[l], f0, g.w[l + 1], g.h[l + 1], f1, 2.0f, stream, pairs,
                                          cells[l], cells[l] * 3, cells[l + 1], cells[l]));
        else
            NUS_HIP(launch_flow_upsample(f0, g.w[l + 1], g.h[l + 1], f1, g.w[l], g.h[l], 2.0f, stream, pairs, cells[l + 1], cells[l]));
        float *t = f0;
        f0 = f1;
        f1 = t;
        if (refine_iters > 0 && (rc = iterate((uint32_t)l, refine_iters, false, false)) != kOk) return rc;
    }
    // Where the level's final flow is now, and in which format.  (After a launch that warped without storing its flow, f0 names a
    // buffer nothing was written to: nobody reads it.)
    const void *final_flow = f0;
    const bool unstored = warped && d_mid && !out;
    if (!flow_half) {
        if (out && f0 != out && !unstored) NUS_HIP(hipMemcpyAsync(out, f0, cells[0] * pairs * 8, hipMemcpyDeviceToDevice, stream));
        if (out) final_flow = out;
    } else if (wrote_half) { // halves, in the caller's buffer if there is one (the launch wrote there), else in the workspace
        if (out && f0 != out && !unstored) NUS_HIP(hipMemcpyAsync(out, f0, cells[0] * pairs * 4, hipMemcpyDeviceToDevice, stream));
        if (out) final_flow = out;
    } else if (!unstored) { // 2 x f32 per cell in the workspace (solver_out was null): converted into the caller's buffer, or beside it
        void *dst = out;
        if (!dst) {
            if ((rc = reserve(cells[0] * pairs * 4, 10)) != kOk) return rc;
            dst = slot_[10];
        }
        NUS_HIP(launch_flow_to_half(f0, dst, cells[0] * pairs, stream));
        final_flow = dst;
    }
    if (d_mid && !warped) { // the warp kernel behind the estimator, on the flow where it is (the caller's buffer, or the workspace)
        WarpLaunch W;
        W.a = d_frames;
        W.b = d_frames + cells[0] * 4;
        W.a_stride = W.b_stride = cells[0] * 4;
        W.flow = static_cast<const float *>(final_flow);
        W.flow_half = flow_half;
        W.fma = true;
        W.out = d_mid;
        W.w = g.w[0], W.h = g.h[0], W.t = t, W.n_pairs = pairs, W.stream = stream;
        NUS_HIP(launch_warp_blend(W));
    }
    return kOk;
}

int HipFlowEstimator::estimate(const uint8_t *a, const uint8_t *b, uint32_t w, uint32_t h, uint32_t levels,
                               uint32_t coarse_iters, uint32_t refine_iters, float lambda, float *flow_out)
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        CHECK_DIMS(w, h);
        if (!a || !b || !flow_out) return fail(kInvalidArgument, "flow: null pointer");
        int rc = ensure_device();
        if (rc != kOk) return rc;
        const size_t fbytes = (size_t)w * h * 4;
        if ((rc = reserve(fbytes, 6)) != kOk || (rc = reserve(fbytes > (size_t)w * h * 8 ? fbytes : (size_t)w * h * 8, 7)) != kOk) return rc;
        NUS_XFER(upload(slot_[6], a, fbytes, stream_));
        NUS_XFER(upload(slot_[7], b, fbytes, stream_));
    }
    // slot 7 doubles as the flow output once frame B has been converted (estimate_device copies
    // into it last, after every reader of frame B has been enqueued on the same stream)
    int rc = estimate_device(slot_[6], slot_[7], w, h, levels, coarse_iters, refine_iters, lambda, slot_[7], stream_);
    if (rc != kOk) return rc;
    std::lock_guard<std::mutex> lk(mu_);
    NUS_XFER(download(flow_out, slot_[7], (size_t)w * h * 8, stream_));
    NUS_HIP(hipStreamSynchronize(stream_));
    return kOk;
}

} // namespace nus

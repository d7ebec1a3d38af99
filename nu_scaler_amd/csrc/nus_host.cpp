// nus_host.cpp -- HipUpscaler and the factory: the host side of the upscaling path above the gfx950 kernels.
// See nus_host.hpp.  (HipFrameInterpolator: nus_host_interp.cpp.)
#include "nus_host.hpp"

#include <condition_variable>
#include <thread>

#include "nus_copy.hpp"
#include "nus_host_util.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace nus {

namespace {
thread_local std::string g_thread_error;
} // namespace

void set_thread_error(const std::string &msg) { g_thread_error = msg; }
const char *thread_error() { return g_thread_error.c_str(); }

// ---------------------------------------------------------------------------------
// HipUpscaler
// ---------------------------------------------------------------------------------

HipUpscaler::HipUpscaler(Quality q, Algorithm a) : quality_(q), algorithm_(a) {}

HipUpscaler::~HipUpscaler() { release(); }

int HipUpscaler::fail(int status, const std::string &msg)
{
    error_ = msg;
    set_thread_error(msg);
    return status;
}

int HipUpscaler::fail_hip(hipError_t e, const char *what)
{
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? kOutOfMemory : kHipError,
                fmt("HIP error in %s: %s", what, hipGetErrorString(e)));
}

const char *HipUpscaler::name() const
{
    // upscale/mod.rs:1060-1066 for the two reference algorithms
    switch (algorithm_) {
    case Algorithm::Bilinear: return "WgpuBilinearUpscaler";
    case Algorithm::Lanczos3: return "HipLanczos3Upscaler";
    case Algorithm::Bicubic: return "HipBicubicUpscaler";
    case Algorithm::Triangle: return "HipTriangleUpscaler";
    case Algorithm::Fsr1: return "HipFsr1Upscaler";
    case Algorithm::FsrEasu: return "HipFsrEasuUpscaler";
    case Algorithm::FsrRcas: return "HipFsrRcasUpscaler";
    default: return "WgpuNearestUpscaler";
    }
}

ResizeFilter HipUpscaler::resize_filter() const
{
    switch (algorithm_) {
    case Algorithm::Bicubic: return ResizeFilter::CatmullRom;
    case Algorithm::Triangle: return ResizeFilter::Triangle;
    default: return ResizeFilter::Lanczos3;
    }
}

int HipUpscaler::set_quality(Quality q)
{
    std::lock_guard<std::mutex> lk(mu_);
    // quality never changes the arithmetic of the reference's own algorithms
    // (upscale/mod.rs:1072-1077); the FSR1-style passes take their default sharpness from it
    quality_ = q;
    return kOk;
}

int HipUpscaler::set_device(int device)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (initialized_) return fail(kInvalidArgument, "set_device must be called before initialize");
    if (device < 0) return fail(kInvalidArgument, "negative device index");
    device_ = device;
    return kOk;
}

int HipUpscaler::set_bilinear_variant(int variant)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (variant != 0 && variant != 1) return fail(kInvalidArgument, "unknown bilinear variant");
    if (initialized_) return fail(kInvalidArgument, "set_bilinear_variant must be called before initialize");
    wgsl_bilinear_ = variant == 1;
    return kOk;
}

int HipUpscaler::set_lanczos_mode(int mode)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (mode != 0 && mode != 1) return fail(kInvalidArgument, "unknown lanczos mode");
    lanczos_exact_ = mode == 1;
    return kOk;
}

int HipUpscaler::set_option(const char *key, int64_t value)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!key) return fail(kInvalidArgument, "null option key");
    if (!strcmp(key, "force_general")) {
        if (initialized_) return fail(kInvalidArgument, "force_general must be set before initialize");
        force_general_ = value != 0;
        return kOk;
    }
    if (!strcmp(key, "force_per_pixel")) {
        if (initialized_) return fail(kInvalidArgument, "force_per_pixel must be set before initialize");
        force_per_pixel_ = value != 0;
        return kOk;
    }
    if (!strcmp(key, "force_rows")) {
        if (initialized_) return fail(kInvalidArgument, "force_rows must be set before initialize");
        force_rows_ = value != 0;
        return kOk;
    }
    if (!strcmp(key, "single_bands")) { // 1: a frame alone in the pipeline goes through it in row bands (default), 0: whole
        if (value != 0 && value != 1) return fail(kInvalidArgument, "single_bands must be 0 or 1");
        single_bands_ = (int)value;
        return kOk;
    }
    if (!strcmp(key, "single_out_plan")) { // how upscale() cuts a pageable output frame into D2H pieces (plan_chunks)
        if (value < 0 || value > 3) return fail(kInvalidArgument, "single_out_plan out of range");
        single_out_plan_ = (int)value;
        return kOk;
    }
    if (!strcmp(key, "batch_out_chunks")) { // D2H pieces per pageable output frame of upscale_batch / the stream ring
        if (value < 1 || value > kOutChunks) return fail(kInvalidArgument, "batch_out_chunks out of range");
        batch_out_chunks_ = (int)value;
        return kOk;
    }
    if (!strcmp(key, "down_seg_width")) { // output columns per wave of the down-scaling kernel; 0 = the host's choice
        if (initialized_) return fail(kInvalidArgument, "down_seg_width must be set before initialize");
        if (value < 0 || value > 64) return fail(kInvalidArgument, "down_seg_width out of range");
        down_seg_width_ = (uint32_t)value;
        return kOk;
    }
    if (!strcmp(key, "rows_per_wave")) {
        if (value < 0 || value > 4096) return fail(kInvalidArgument, "rows_per_wave out of range");
        rows_per_wave_ = (uint32_t)value;
        return kOk;
    }
    if (!strcmp(key, "edge_stream")) { // 1: the x2 edge-column pass of a batch runs beside the main kernel on a second stream
        if (value != 0 && value != 1) return fail(kInvalidArgument, "edge_stream must be 0 or 1");
        edge_stream_ = (int)value;
        return kOk;
    }
    if (!strcmp(key, "unit_order")) { // wave order of upscale_unit_device: 0 frame-major, 1 row-block-major (default)
        if (value != 0 && value != 1) return fail(kInvalidArgument, "unit_order must be 0 or 1");
        unit_order_ = (uint32_t)value;
        return kOk;
    }
    if (!strcmp(key, "fsr_two_pass")) { // FSR1: EASU -> scratch image -> row-walking RCAS for frames of >= 1 MiB (default 1); 0 = fused tile always
        if (initialized_) return fail(kInvalidArgument, "fsr_two_pass must be set before initialize");
        if (value != 0 && value != 1) return fail(kInvalidArgument, "fsr_two_pass must be 0 or 1");
        fsr_two_pass_ = value != 0;
        return kOk;
    }
    if (!strcmp(key, "fsr_fast")) { // FSR1-style EASU in FAST arithmetic (nus_k_fsr.hip): 0 = the shaders' own operation order (default)
        if (value != 0 && value != 1) return fail(kInvalidArgument, "fsr_fast must be 0 or 1");
        fsr_fast_ = value != 0;
        return kOk;
    }
    if (!strcmp(key, "pq_narrow")) { // 1 (default): the x P/Q kernel runs a support-2 filter's sums over its 4 non-zero taps; 0: all 6 slots
        if (value != 0 && value != 1) return fail(kInvalidArgument, "pq_narrow must be 0 or 1");
        pq_narrow_allowed_ = value != 0;
        return kOk;
    }
    if (!strcmp(key, "inject_retire_error")) { // TEST HOOK: the value-th frame retired from now on (1 = the next) reports a HIP error
        // refused unless the PROCESS asked for test hooks (environment NUS_TEST_HOOKS=1, read per call): not something a caller of
        // the production library can switch on by accident; release() clears it
        const char *hooks = getenv("NUS_TEST_HOOKS");
        if (!hooks || strcmp(hooks, "1") != 0) return fail(kInvalidArgument, "unknown option 'inject_retire_error'");
        if (value < 0 || value > 1000000) return fail(kInvalidArgument, "inject_retire_error out of range");
        inject_retire_.store((int)value);
        return kOk;
    }
    return fail(kInvalidArgument, fmt("unknown option '%s'", key));
}

int HipUpscaler::get_option(const char *key, int64_t *value)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!key || !value) return fail(kInvalidArgument, "null option key or result pointer");
    const bool pq = initialized_ && variant_ == Variant::LanczosPqRegWin;
    if (!strcmp(key, "pq_p")) *value = pq ? pq_p_ : 0;
    else if (!strcmp(key, "pq_q")) *value = pq ? pq_q_ : 0;
    else if (!strcmp(key, "pq_narrow_active")) *value = pq && pq_narrow_ && pq_narrow_allowed_ ? 1 : 0;
    else if (!strcmp(key, "rows_per_wave")) *value = rows_per_wave_;
    else return fail(kInvalidArgument, fmt("unknown option '%s'", key));
    return kOk;
}

int HipUpscaler::set_input_format(int format)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (format < 0 || format > 3) return fail(kInvalidArgument, "unknown input format");
    in_format_ = format;
    return kOk;
}

int HipUpscaler::set_sharpness(float easu, float rcas)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (easu > 1.0f || rcas > 1.0f || easu != easu || rcas != rcas)
        return fail(kInvalidArgument, "sharpness must be <= 1 (negative: quality default)");
    easu_sharp_ = easu;
    rcas_sharp_ = rcas;
    return kOk;
}

// EASU's "quality-dependent sharpness factor" (fsr.rs:35) is never given a value by the
// reference; build-defined default 0 (no pull toward the centre texel).
float HipUpscaler::easu_sharpness() const { return easu_sharp_ >= 0.0f ? easu_sharp_ : 0.0f; }

// Default per quality as the reference's CPU FSR path sets it (Nu_scale/src/upscale/fsr3.rs:231-236).
float HipUpscaler::rcas_sharpness() const
{
    if (rcas_sharp_ >= 0.0f) return rcas_sharp_;
    switch (quality_) {
    case Quality::Ultra: return 0.8f;
    case Quality::Quality: return 0.7f;
    case Quality::Balanced: return 0.6f;
    default: return 0.5f;
    }
}

int HipUpscaler::ensure_device()
{
    const int n = device_count();
    if (n <= 0)
        return fail(kNoDevice, "no HIP device available (the gfx950 path has no CPU fallback)");
    if (device_ >= n) return fail(kNoDevice, fmt("HIP device %d requested but only %d present", device_, n));
    NUS_HIP(hipSetDevice(device_));
    return kOk;
}

void HipUpscaler::release()
{
    if (ring_.open) { // (initialize() again or the destructor with a stream still open: stop its thread first)
        {
            std::lock_guard<std::mutex> rl(ring_.m);
            ring_.stop = true;
        }
        ring_.cv.notify_all();
        if (ring_.thread.joinable()) ring_.thread.join();
        ring_.open = false;
    }
    inject_retire_.store(0); // (the test hook does not outlive an initialize())
    if (device_count() > 0) (void)hipSetDevice(device_);
    for (void *p : table_allocs_) (void)hipFree(p);
    table_allocs_.clear();
    dt_ = DeviceTables();
    for (hipEvent_t ev : prof_events_) (void)hipEventDestroy(ev);
    prof_events_.clear();
    prof_used_ = 0;
    for (Slot &s : slots_) release_slot(s);
    for (hipStream_t *st : {&s_in_, &s_k_, &s_out_, &s_edge_}) {
        if (*st) (void)hipStreamDestroy(*st);
        *st = nullptr;
    }
    if (fsr_scratch_) (void)hipFree(fsr_scratch_);
    fsr_scratch_ = nullptr;
    for (hipEvent_t *ev : {&ev_fork_, &ev_join_, &ev_fsr_}) {
        if (*ev) (void)hipEventDestroy(*ev);
        *ev = nullptr;
    }
    initialized_ = false;
    have_ms_ = false;
}

// widest input-column footprint of a `segw`-column output segment (what one wave's LDS row / register columns must hold)
uint32_t HipUpscaler::widest_footprint(uint32_t segw) const
{
    uint32_t widest = 0;
    for (uint32_t x0 = 0; x0 < ow_; x0 += segw) {
        const uint32_t xl = (x0 + segw < ow_ ? x0 + segw : ow_) - 1;
        const uint32_t span = (uint32_t)(tx_.lz_left[xl] + (int32_t)tx_.lz_ntaps[xl] - tx_.lz_left[x0]);
        if (span > widest) widest = span;
    }
    return widest;
}

// widest union of the tap windows of `n` adjacent outputs (a lane's outputs in the union-window H pass); ow_ % n == 0.  For two
// outputs per lane a window is as long as the axis's widest one (7 for Lanczos-3 on an up-scale; the table's slots beyond a window's
// taps hold zeros), so that their union fits the 8-tap pass; for four the table's 8 slots as before (counted tightly, x1.4 - x1.5 would
// move from the plain 8-slot pass to the 10-tap union with its weights in LDS, which is slower there: 768p -> 1080p 6.1 -> 7.3 us)
uint32_t HipUpscaler::widest_union(uint32_t n) const
{
    uint32_t widest = 0;
    const uint32_t taps = n == 2 && tx_.lz_max_taps > 0 && tx_.lz_max_taps < 8 ? (uint32_t)tx_.lz_max_taps : 8u;
    for (uint32_t x0 = 0; x0 < ow_; x0 += n) {
        const uint32_t u = (uint32_t)(tx_.lz_left[x0 + n - 1] - tx_.lz_left[x0]) + taps;
        if (u > widest) widest = u;
    }
    return widest;
}

// Kernel for the resize filters (Lanczos-3 / Catmull-Rom / Triangle), most specialised first.
void HipUpscaler::choose_resize_variant(bool x2)
{
    variant_ = Variant::LanczosGeneral; // per-pixel fallback
    xs_factor_ = 0;
    resize_ncols_max_ = resize_union_taps_ = 0;
    resize_small_taps_ = false;
    win_outputs_per_lane_ = 4;
    const bool addressable = (uint64_t)ow_ * oh_ * 4 < (1ull << 31); // buffer-resource addressing of the output frame
    // exact x2: fixed interior weights; same ratio on both axes, so the two passes share the same 12 numbers
    if (x2 && iw_ >= 16 && ih_ >= 16 && addressable && lanczos_x2_phase_frame(tx_, wx6_) && lanczos_x2_phase_frame(ty_, wy6_) &&
        lanczos_x2_interior_uniform(tx_, wx6_) && lanczos_x2_interior_uniform(ty_, wy6_) &&
        memcmp(&wx6_[(size_t)8 * 6], &wy6_[(size_t)8 * 6], 12 * sizeof(float)) == 0) {
        variant_ = Variant::LanczosX2RegWin;
        return;
    }
    // integer factors x3 / x4: the same register-window design (nus_k_lanczos_xs.hip).  x4: the interior weights are
    // uniform (one set per phase, the same numbers on both axes).  x3: they move with the binade of the coordinate
    // (nus_tables.hpp), so every lane / row takes the set of its class
    xs_cls_x_.clear(), xs_cls_y_.clear(), xs_wcls_x_.clear(), xs_wcls_y_.clear();
    for (uint32_t S = 3; S <= 4 && !force_general_; ++S) {
        if (ow_ != S * iw_ || oh_ != S * ih_ || (iw_ % 4) != 0 || iw_ < 16 || ih_ < 16 || !addressable) continue;
        if (!lanczos_xs_phase_frame(tx_, S, wx6_) || !lanczos_xs_phase_frame(ty_, S, wy6_)) continue;
        if (lanczos_xs_interior_uniform(tx_, S, wx6_) && lanczos_xs_interior_uniform(ty_, S, wy6_) &&
            memcmp(&wx6_[(size_t)8 * S * 6], &wy6_[(size_t)8 * S * 6], (size_t)S * 6 * sizeof(float)) == 0) {
            xs_factor_ = S;
            variant_ = Variant::LanczosXsRegWin;
            return;
        }
        if (S == 3 && lanczos_xs_weight_classes(tx_, S, wx6_, true, xs_cls_x_, xs_wcls_x_) &&
            lanczos_xs_weight_classes(ty_, S, wy6_, false, xs_cls_y_, xs_wcls_y_)) {
            xs_factor_ = S;
            variant_ = Variant::LanczosXsRegWin;
            return;
        }
        xs_cls_x_.clear(), xs_cls_y_.clear(), xs_wcls_x_.clear(), xs_wcls_y_.clear();
    }
    // x3/2 (720p -> 1080p, 1440p -> 4K): the same design with three output rows per input row pair (nus_k_lanczos_r32.hip);
    // weights by class of the input pair on both axes, as at x3
    if (!force_general_ && 2 * (uint64_t)ow_ == 3 * (uint64_t)iw_ && 2 * (uint64_t)oh_ == 3 * (uint64_t)ih_ && (iw_ % 8) == 0 &&
        (ih_ % 2) == 0 && iw_ >= 32 && ih_ >= 16 && addressable && lanczos_r32_phase_frame(tx_, wx6_) &&
        lanczos_r32_phase_frame(ty_, wy6_) && lanczos_r32_weight_classes(tx_, wx6_, true, xs_cls_x_, xs_wcls_x_) &&
        lanczos_r32_weight_classes(ty_, wy6_, false, xs_cls_y_, xs_wcls_y_)) {
        variant_ = Variant::LanczosR32RegWin;
        return;
    }
    xs_cls_x_.clear(), xs_cls_y_.clear(), xs_wcls_x_.clear(), xs_wcls_y_.clear();
    // x4/3 (1080p -> 1440p): four output rows per group of three input rows (nus_k_lanczos_r43.hip); the ratio 3/4 is exact in
    // f32, so the interior weights are uniform (one set per phase, the same numbers on both axes), as at x2 and x4
    if (!force_general_ && 3 * (uint64_t)ow_ == 4 * (uint64_t)iw_ && 3 * (uint64_t)oh_ == 4 * (uint64_t)ih_ && (iw_ % 12) == 0 &&
        (ih_ % 3) == 0 && iw_ >= 48 && ih_ >= 18 && addressable && lanczos_r43_phase_frame(tx_, wx6_) &&
        lanczos_r43_phase_frame(ty_, wy6_) && lanczos_r43_interior_uniform(tx_, wx6_) && lanczos_r43_interior_uniform(ty_, wy6_) &&
        memcmp(&wx6_[(size_t)8 * 6], &wy6_[(size_t)8 * 6], 24 * sizeof(float)) == 0) {
        variant_ = Variant::LanczosR43RegWin;
        return;
    }
    // x5/4, x6/5, x5/3, x5/2: P output rows per group of Q input rows, every output's weights from the tables in frame form
    // (nus_k_lanczos_pq.hip); the ratios are not exact in f32, so nothing about the weights is assumed beyond their frames
    pq_p_ = pq_q_ = 0;
    pq_narrow_ = false;
    if (!force_general_ && ow_ > iw_ && addressable && iw_ >= 32 && ih_ >= 12) {
        uint64_t a = ow_, b = iw_;
        while (b) {
            const uint64_t r = a % b;
            a = b, b = r;
        }
        const uint32_t P = (uint32_t)(ow_ / a), Q = (uint32_t)(iw_ / a);
        if (lanczos_pq_supported(P, Q) && (uint64_t)oh_ * Q == (uint64_t)ih_ * P && (iw_ % Q) == 0 && (ih_ % Q) == 0 && (ow_ % 4) == 0 &&
            lanczos_pq_phase_frame(tx_, P, Q, wx6_) && lanczos_pq_phase_frame(ty_, P, Q, wy6_)) {
            pq_p_ = P, pq_q_ = Q;
            // a filter of support 2 or less (Catmull-Rom, Triangle): slots 0 and 5 of every output's frame are zero on both axes, and
            // the kernel may sum slots 1 .. 4 only (checked on the tables themselves, border outputs included)
            auto edge_slots_zero = [](const std::vector<float> &w6) {
                for (size_t o = 0; o + 6 <= w6.size(); o += 6)
                    if (w6[o] != 0.0f || w6[o + 5] != 0.0f) return false;
                return true;
            };
            pq_narrow_ = edge_slots_zero(wx6_) && edge_slots_zero(wy6_);
        }
    }
    if (pq_p_) {
        // The any-scale kernel's parameters as well: at x6/5, x8/5, x9/5 the kernel's EXACT instantiations hold 250 - 430 registers (one
        // wave per SIMD) and lose to it (1080p: 11.2 / 25.4 / 30.0 against 8.3 / 19.9 / 28.5 us; x7/5 wins, 11.6 against 17.4), so EXACT
        // mode -- which can be switched on after initialize() -- takes it there (enqueue)
        choose_general_resize_variant();
        pq_exact_fallback_ = variant_ == Variant::ResizeWin && pq_q_ == 5 && pq_p_ != 7;
        variant_ = Variant::LanczosPqRegWin;
        return;
    }
    choose_general_resize_variant();
}

// ... and the kernels for every other shape: streamed down-scaling, the register-window / LDS-row any-scale kernels, per pixel
void HipUpscaler::choose_general_resize_variant()
{
    variant_ = Variant::LanczosGeneral;
    if (force_per_pixel_) return;
    // vertical down-scaling: stream the input rows through 7 accumulator slots, if the windows allow it and a
    // 64-column output segment's footprint fits 5 columns per lane
    if (ih_ > oh_ && tx_.lz_max_taps <= 32 && !force_rows_) {
        const uint32_t widest64 = widest_footprint(64);
        if (build_down_stream_tables(ty_, down_rows_, down_done_) && widest64 <= 320) {
            variant_ = Variant::ResizeDown;
            // Output columns per wave: every lane carries ceil(footprint / 64) input columns through the vertical pass whether
            // the segment fills them or not, so a slightly narrower segment whose footprint fits one column per lane fewer is
            // cheaper (2x down: 58 outputs over 128 columns instead of 64 over 140 = 3 columns per lane, a third of them idle).
            // Cost per output column ~ (vertical FMAs of a lane per output row + its horizontal pass: an LDS read and 4 FMAs per
            // tap of the compiled trip count, priced at 10) / width, calibrated on profiles/r03_resize_down_segment_width_ab.txt
            // (4K -> 1080p: 58 columns 16.9 us against 20.1 at 64; 4K -> 720p, 32-tap passes: 64 stays, 57 is 4 % slower).
            down_seg_w_ = 64;
            resize_ncols_max_ = widest64;
            if (down_seg_width_ == 0) {
                const double rows_per_out = (double)ih_ / oh_;
                double best = 0.0;
                for (uint32_t w = 64; w >= 40; --w) {
                    const uint32_t fp = widest_footprint(w);
                    const double cost = (28.0 * ((fp + 63) / 64) * rows_per_out + 10.0 * (tx_.lz_max_taps > 16 ? 32 : 16) + 30.0) / w;
                    if (w == 64 || cost < best * 0.97) best = cost, down_seg_w_ = w, resize_ncols_max_ = fp;
                }
            } else { // option "down_seg_width" (tests, A/B timing)
                down_seg_w_ = down_seg_width_ > 64 ? 64 : down_seg_width_;
                resize_ncols_max_ = widest_footprint(down_seg_w_);
                if (resize_ncols_max_ > 320) down_seg_w_ = 64, resize_ncols_max_ = widest64;
            }
            return;
        }
    }
    // LDS row kernel: the widest segment footprint must fit the per-wave LDS row (4 waves x 640 x 16 B = 40 KiB per block)
    const uint32_t widest = widest_footprint((ow_ % 4) == 0 ? 256 : 64);
    if (widest > 640) return;
    variant_ = Variant::ResizeRows;
    resize_ncols_max_ = widest;
    resize_small_taps_ = tx_.lz_max_taps <= 8 && ty_.lz_max_taps <= 8;
    if ((ow_ % 4) != 0 || !resize_small_taps_) return;
    resize_union_taps_ = widest_union(4); // union-window H pass, 4 outputs per lane
    // register-window variant: up-scaling shapes whose first tap row moves by at most one per output row
    bool win_ok = ty_.lz_max_taps <= 7 && !force_rows_;
    for (uint32_t y = 1; win_ok && y < oh_; ++y) {
        const int32_t d = ty_.lz_left[y] - ty_.lz_left[y - 1];
        win_ok = d == 0 || d == 1;
    }
    if (!win_ok) return;
    // Four outputs per lane (segments of 256 output columns) from x2 up.  Below x2 two outputs per lane win wherever their 8-tap union
    // pass applies (two input columns per lane, union weights in registers, rows through the ring -- the four-output shape there needs
    // three columns per lane and keeps its union weights in LDS): 768p -> 1080p 7.2 -> 6.5, x1.7 18.6 -> 15.8, x1.9 21.2 -> 19.0 us per
    // frame; from x2.2 up the four-output shape is the faster one again (profiles/r05_resize_win_rotating_window.txt)
    const bool two_per_lane = 2 * (uint64_t)iw_ > ow_ && widest_footprint(128) <= 128 && widest_union(2) <= 8;
    if (widest <= 192 && !two_per_lane) {
        variant_ = Variant::ResizeWin;
        return;
    }
    // factors x1.0 .. x2: two outputs per lane, segments of 128 output columns
    const uint32_t widest2 = widest_footprint(128);
    if (widest2 > 192) return;
    variant_ = Variant::ResizeWin;
    win_outputs_per_lane_ = 2;
    resize_ncols_max_ = widest2;
    resize_union_taps_ = widest_union(2);
}

// One of the small rational factors P/Q the fixed-ratio nearest / bilinear kernels are built for, with tables of exactly the
// shape they assume on both axes: source index Q (o / P) + (o % P) Q / P and, for bilinear, fraction 0 where the phase lands
// on a pixel ((o % P) Q % P == 0).
bool HipUpscaler::ratio_shape(bool bilinear) const
{
    static const uint32_t kRatios[][2] = {{3, 2}, {4, 3}, {3, 1}, {4, 1}, {2, 1}, {5, 4}, {6, 5}, {5, 3}, {5, 2}, {7, 2}, {7, 5}};
    for (const auto &r : kRatios) { // (x8/5: the table kernel is the faster one for nearest too, 5.6 against 5.8 us at 1080p)
        const uint32_t P = r[0], Q = r[1];
        if ((uint64_t)ow_ * Q != (uint64_t)iw_ * P || (uint64_t)oh_ * Q != (uint64_t)ih_ * P || iw_ % Q != 0 || ih_ % Q != 0) continue;
        // bilinear at x5/2 and x7/2: five / seven lerped outputs from two input columns per lane leave the fixed-ratio kernel behind the
        // table kernel (864p -> 4K: 11.7 against 10.6 us per frame, profiles/r05_nearest_bilinear_pq_ratios.txt); nearest gains at every factor
        if (bilinear && Q == 2 && P >= 5) return false;
        if (bilinear && Q == 5 && P >= 7) return false; // (nearest only: Q + 1 lerped rows of 4 P values do not fit the registers)
        bool ok = true;
        for (const AxisTables *t : {&tx_, &ty_})
            for (uint32_t o = 0; ok && o < t->out_n; ++o) {
                const uint32_t want = Q * (o / P) + (o % P) * Q / P;
                ok = bilinear ? (t->bl_i0[o] == want && (((o % P) * Q) % P != 0 || t->bl_frac[o] == 0.0f)) : t->nn_src[o] == want;
            }
        return ok;
    }
    return false;
}

void HipUpscaler::choose_variant()
{
    const bool x2 = ow_ == 2 * iw_ && oh_ == 2 * ih_ && (iw_ % 4) == 0 && !force_general_;
    switch (algorithm_) {
    case Algorithm::Nearest:
        variant_ = x2 ? Variant::NearestX2 : Variant::NearestTable;
        if (!x2 && !force_general_ && ratio_shape(false)) variant_ = Variant::NearestRatio;
        break;
    case Algorithm::Bilinear: {
        // The packed-integer x2 kernel is valid iff the CPU-form tables are exactly
        // i0 = o>>1, frac = 0 / 0.5 (and frac 0 at the clamped last sample).
        bool ok = x2 && !wgsl_bilinear_;
        for (const AxisTables *t : {&tx_, &ty_}) {
            for (uint32_t o = 0; ok && o < t->out_n; ++o) {
                const bool last_odd = (o & 1) && (o >> 1) == t->in_n - 1;
                const float want = (o & 1) && !last_odd ? 0.5f : 0.0f;
                ok = t->bl_i0[o] == (o >> 1) && t->bl_frac[o] == want;
            }
        }
        variant_ = ok ? Variant::BilinearX2Int : Variant::BilinearTable;
        if (!ok && !wgsl_bilinear_ && !force_general_ && ratio_shape(true)) variant_ = Variant::BilinearRatio;
        break;
    }
    case Algorithm::Lanczos3:
    case Algorithm::Bicubic:
    case Algorithm::Triangle: choose_resize_variant(x2); break;
    case Algorithm::Fsr1:
        // two kernels with the EASU image in HBM between them beat the fused LDS tile on real frames (1080p -> 4K: EASU 65.5 + RCAS rows
        // 21.7 against 98.2 us fused; FAST 43.0 + 21.7 against 80.2): the tile's phases add up, the two launches overlap their own
        variant_ = fsr_two_pass_ && (size_t)ow_ * oh_ * 4 >= ((size_t)1 << 20) ? Variant::Fsr1TwoPass : Variant::Fsr1Fused;
        break;
    case Algorithm::FsrEasu: variant_ = Variant::FsrEasu; break;
    case Algorithm::FsrRcas: variant_ = Variant::FsrRcas; break;
    }
}

int HipUpscaler::upload_tables()
{
    for (void *p : table_allocs_) (void)hipFree(p);
    table_allocs_.clear();
    dt_ = DeviceTables();
    auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
        void *d = nullptr;
        NUS_HIP(hipMalloc(&d, bytes));
        table_allocs_.push_back(d);
        NUS_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *dst = d;
        return kOk;
    };
    int rc = kOk;
#define UP(vec, field)                                                                                   \
    if (rc == kOk) rc = up((vec).data(), (vec).size() * sizeof((vec)[0]), reinterpret_cast<const void **>(&dt_.field))
    switch (algorithm_) {
    case Algorithm::Nearest:
        UP(tx_.nn_src, nn_sx);
        UP(ty_.nn_src, nn_sy);
        break;
    case Algorithm::Bilinear:
        UP(tx_.bl_i0, bl_x0);
        UP(tx_.bl_frac, bl_fx);
        UP(ty_.bl_i0, bl_y0);
        UP(ty_.bl_frac, bl_fy);
        break;
    case Algorithm::Lanczos3:
    case Algorithm::Bicubic:
    case Algorithm::Triangle:
        UP(tx_.lz_left, lz_lx);
        UP(tx_.lz_ntaps, lz_nx);
        UP(tx_.lz_w, lz_wx);
        UP(ty_.lz_left, lz_ly);
        UP(ty_.lz_ntaps, lz_ny);
        UP(ty_.lz_w, lz_wy);
        dt_.lz_stride = kResizeMaxTaps;
        if (variant_ == Variant::ResizeDown) {
            UP(down_rows_, lz_down_rows);
            UP(down_done_, lz_down_done);
        }
        if (variant_ == Variant::LanczosXsRegWin) {
            UP(wy6_, lz_wy6);
            if (!xs_cls_x_.empty()) {
                UP(xs_cls_x_, lz_xs_cls_x);
                UP(xs_cls_y_, lz_xs_cls_y);
                UP(xs_wcls_x_, lz_xs_wcls_x);
                UP(xs_wcls_y_, lz_xs_wcls_y);
            }
            for (uint32_t p = 0; p < xs_factor_; ++p)
                for (int j = 0; j < 6; ++j) dt_.lz_wxs[p][j] = wx6_[((size_t)8 * xs_factor_ + p) * 6 + j];
            for (uint32_t q = 0; q < 4 * xs_factor_; ++q)
                for (int j = 0; j < 6; ++j) {
                    dt_.lz_wxs_left[q][j] = wx6_[(size_t)q * 6 + j];
                    dt_.lz_wxs_right[q][j] = wx6_[((size_t)ow_ - 4 * xs_factor_ + q) * 6 + j];
                }
        }
        if (variant_ == Variant::LanczosR43RegWin) {
            for (uint32_t p = 0; p < 4; ++p)
                for (int j = 0; j < 6; ++j) dt_.lz_wxs[p][j] = wx6_[((size_t)8 + p) * 6 + j];
            for (uint32_t q = 0; q < 8; ++q)
                for (int j = 0; j < 6; ++j) {
                    dt_.lz_wxs_left[q][j] = wx6_[(size_t)q * 6 + j];
                    dt_.lz_wxs_right[q][j] = wx6_[((size_t)ow_ - 8 + q) * 6 + j];
                }
            UP(wy6_, lz_wy6);
        }
        if (variant_ == Variant::LanczosPqRegWin) {
            UP(wx6_, lz_wx6);
            UP(wy6_, lz_wy6);
        }
        if (variant_ == Variant::LanczosR32RegWin) {
            for (uint32_t q = 0; q < 12; ++q)
                for (int j = 0; j < 6; ++j) {
                    dt_.lz_wxs_left[q][j] = wx6_[(size_t)q * 6 + j];
                    dt_.lz_wxs_right[q][j] = wx6_[((size_t)ow_ - 12 + q) * 6 + j];
                }
            UP(wy6_, lz_wy6);
            UP(xs_cls_x_, lz_xs_cls_x);
            UP(xs_cls_y_, lz_xs_cls_y);
            UP(xs_wcls_x_, lz_xs_wcls_x);
            UP(xs_wcls_y_, lz_xs_wcls_y);
        }
        if (variant_ == Variant::LanczosX2RegWin) {
            UP(wy6_, lz_wy6);
            for (int j = 0; j < 6; ++j) {
                dt_.lz_wxe[j] = wx6_[(size_t)8 * 6 + j];
                dt_.lz_wxo[j] = wx6_[(size_t)9 * 6 + j];
            }
            for (int i = 0; i < 48; ++i) {
                dt_.lz_wx_left[i] = wx6_[i];
                dt_.lz_wx_right[i] = wx6_[(size_t)(ow_ - 8) * 6 + i];
            }
        }
        break;
    default: break; // FSR1-style passes compute their coordinates in the kernel, as the shaders do
    }
#undef UP
    return rc;
}

int HipUpscaler::initialize(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (in_w == 0 || in_h == 0 || out_w == 0 || out_h == 0)
        return fail(kInvalidArgument, "initialize: dimensions must be non-zero");
    const uint64_t lim = 1ull << 31;
    if ((uint64_t)in_w * in_h * 4 >= lim * 4 || (uint64_t)out_w * out_h * 4 >= lim * 4 || in_w > (1u << 20) ||
        in_h > (1u << 20) || out_w > (1u << 20) || out_h > (1u << 20))
        return fail(kInvalidArgument, "initialize: frame too large");
    int rc = ensure_device();
    if (rc != kOk) return rc;
    release(); // any dimension change is a full re-init (upscale/mod.rs:883-889)
    iw_ = in_w;
    ih_ = in_h;
    ow_ = out_w;
    oh_ = out_h;
    build_axis_tables(iw_, ow_, wgsl_bilinear_, tx_, resize_filter());
    build_axis_tables(ih_, oh_, wgsl_bilinear_, ty_, resize_filter());
    if (is_resize() && (tx_.lz_max_taps < 0 || ty_.lz_max_taps < 0))
        return fail(kUnsupported, fmt("Lanczos-3 window exceeds %u taps for %ux%u -> %ux%u", kResizeMaxTaps, iw_, ih_, ow_, oh_));
    if (algorithm_ == Algorithm::FsrRcas && (iw_ != ow_ || ih_ != oh_))
        return fail(kInvalidArgument, "initialize: RCAS alone is a same-size pass (output size must equal input size)");
    choose_variant();
    rc = upload_tables();
    if (rc != kOk) return rc;
    if (variant_ == Variant::Fsr1TwoPass) { // the EASU images between the two passes: allocated here so that enqueue allocates nothing
        NUS_HIP(hipMalloc(reinterpret_cast<void **>(&fsr_scratch_), (size_t)kFsrScratchFrames * ow_ * oh_ * 4));
        NUS_HIP(hipEventCreateWithFlags(&ev_fsr_, hipEventDisableTiming));
        fsr_pending_ = false;
    }
    if (variant_ == Variant::LanczosX2RegWin) { // the edge stream of enqueue(): made here so that enqueue allocates nothing
        NUS_HIP(hipStreamCreateWithFlags(&s_edge_, hipStreamNonBlocking));
        NUS_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        NUS_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    }
    initialized_ = true;
    error_.clear();
    return kOk;
}

// rows per wave of the x2 resize kernel when the caller has not set them: enough waves to fill 256 CUs a few times over, tall
// enough to amortise the 6 halo rows
uint32_t HipUpscaler::lanczos_x2_rows_per_wave(uint32_t n_frames, bool unit) const
{
    if (rows_per_wave_) return rows_per_wave_;
    const uint64_t nstrips = (iw_ + kLanczosX2StripCols - 1) / kLanczosX2StripCols;
    const uint64_t rows_total = (uint64_t)ih_ * nstrips * n_frames * (unit ? 2 : 1);
    const uint64_t t = rows_total / 8192;
    // The one-launch step of a big batch walks 108 rows per wave (10 row blocks per 1080p frame: 6 halo rows per 108 instead of
    // per 36) as long as that leaves a dozen rounds of resident waves: +1.2 % on the 300-unit step
    // (profiles/r04_unit_step_rows_per_wave.txt); the plain kernel gains nothing beyond 36.
    const uint64_t big = unit ? rows_total / (3072 * 12) : 0;
    if (big > 36) return (uint32_t)(big > 108 ? 108 : big);
    return (uint32_t)(t < 8 ? 8 : (t > 36 ? 36 : t));
}

uint32_t HipUpscaler::band_alignment(uint32_t n_frames) const
{
    switch (variant_) {
    case Variant::LanczosX2RegWin: return lanczos_x2_rows_per_wave(n_frames, false);
    case Variant::NearestX2:
    case Variant::BilinearX2Int: return 4;
    default: return 0;
    }
}

int HipUpscaler::enqueue(const uint8_t *d_in, uint8_t *d_out, uint32_t n_frames, hipStream_t stream, const BlendSrc *blend,
                         const UnitDst *unit, uint32_t row0, uint32_t rows)
{
    UpscaleLaunch L;
    L.in = d_in;
    L.out = d_out;
    L.row0 = row0;
    L.rows = rows;
    L.iw = iw_;
    L.ih = ih_;
    L.ow = ow_;
    L.oh = oh_;
    L.n_frames = n_frames;
    L.stream = stream;
    L.in_sel = input_selector(in_format_);
    if (blend) {
        L.in_stride = blend->a_stride;
        L.in_b = blend->b;
        L.in_b_stride = blend->b_stride;
        L.blend_t = blend->t;
    }
    // with profiling on, bracket the frames' launches (main kernel and, for the x2 resize, its edge-column pass)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    if (profiling_) {
        if (prof_used_ + 2 > prof_events_.size()) {
            for (int i = 0; i < 64; ++i) {
                hipEvent_t ev;
                NUS_HIP(hipEventCreate(&ev));
                prof_events_.push_back(ev);
            }
        }
        ev_begin = prof_events_[prof_used_];
        ev_end = prof_events_[prof_used_ + 1];
        prof_used_ += 2;
        NUS_HIP(hipEventRecord(ev_begin, stream));
    }
    // (see below: the x2 resize kernels' edge-column pass of a batch runs beside the main kernel; the fork point is here, in
    // front of the main kernel's launch)
    constexpr uint32_t kEdgeBesideMinFrames = 8; // below that the event operations cost more than the pass
    const bool beside = variant_ == Variant::LanczosX2RegWin && edge_stream_ && s_edge_ && ev_fork_ && ev_join_ &&
                        n_frames >= kEdgeBesideMinFrames;
    if (beside) {
        NUS_HIP(hipEventRecord(ev_fork_, stream));
        NUS_HIP(hipStreamWaitEvent(s_edge_, ev_fork_, 0));
    }
    hipError_t e = hipSuccess;
    switch (variant_) {
    case Variant::NearestTable: e = launch_nearest_table(L, dt_); break;
    case Variant::NearestX2: e = launch_nearest_x2(L); break;
    case Variant::NearestRatio: e = launch_nearest_ratio(L); break;
    case Variant::BilinearTable: e = launch_bilinear_table(L, dt_, wgsl_bilinear_); break;
    case Variant::BilinearX2Int: e = launch_bilinear_x2_int(L); break;
    case Variant::BilinearRatio: e = launch_bilinear_ratio(L, dt_); break;
    case Variant::LanczosGeneral: e = launch_lanczos_general(L, dt_, lanczos_exact_, 0); break;
    case Variant::ResizeWin:
        e = launch_resize_win(L, dt_, lanczos_exact_, resize_ncols_max_, resize_union_taps_, win_outputs_per_lane_);
        break;
    case Variant::ResizeRows:
        e = launch_resize_rows(L, dt_, lanczos_exact_, resize_ncols_max_, resize_small_taps_, resize_union_taps_);
        break;
    case Variant::ResizeDown: e = launch_resize_down(L, dt_, lanczos_exact_, resize_ncols_max_, tx_.lz_max_taps, down_seg_w_); break;
    case Variant::FsrEasu: e = launch_fsr1(L, 0, easu_sharpness(), rcas_sharpness(), fsr_fast_); break;
    case Variant::FsrRcas: e = launch_fsr1(L, 1, easu_sharpness(), rcas_sharpness()); break;
    case Variant::Fsr1Fused: e = launch_fsr1(L, 2, easu_sharpness(), rcas_sharpness(), fsr_fast_); break;
    case Variant::Fsr1TwoPass: {
        // chunks of kFsrScratchFrames frames: EASU into the scratch images, RCAS (rows) out of them.  The scratch belongs to the
        // handle: a batch on another stream waits for the previous batch's last RCAS before it overwrites them.
        if (fsr_pending_) e = hipStreamWaitEvent(stream, ev_fsr_, 0);
        const size_t in_frame = (size_t)iw_ * ih_ * 4, out_frame = (size_t)ow_ * oh_ * 4;
        for (uint32_t done = 0; done < n_frames && e == hipSuccess; done += kFsrScratchFrames) {
            const uint32_t m = n_frames - done < kFsrScratchFrames ? n_frames - done : kFsrScratchFrames;
            UpscaleLaunch E = L;
            E.in = L.in + (size_t)done * in_frame;
            E.out = fsr_scratch_;
            E.n_frames = m;
            e = launch_fsr1(E, 0, easu_sharpness(), rcas_sharpness(), fsr_fast_);
            if (e != hipSuccess) break;
            UpscaleLaunch Rc = L;
            Rc.in = fsr_scratch_;
            Rc.out = L.out + (size_t)done * out_frame;
            Rc.iw = ow_, Rc.ih = oh_;
            Rc.n_frames = m;
            Rc.in_sel = kSelRGBA; // EASU has swizzled already
            e = launch_fsr1(Rc, 1, easu_sharpness(), rcas_sharpness());
        }
        if (e == hipSuccess) e = hipEventRecord(ev_fsr_, stream);
        fsr_pending_ = e == hipSuccess;
        break;
    }
    case Variant::LanczosR43RegWin: {
        uint32_t th = rows_per_wave_;
        if (th == 0) { // as at x3/2: 24 - 36 rows per wave on a batch, equal row blocks (taller blocks: + 5 - 10 %)
            const uint64_t rows_total = (uint64_t)ih_ * ((iw_ + 185) / 186) * n_frames;
            uint64_t t = rows_total / 12288;
            t = t < 12 ? 12 : (t > 36 ? 36 : t);
            const uint64_t blocks = (ih_ + t - 1) / t;
            th = (uint32_t)(((ih_ + blocks - 1) / blocks + 2) / 3 * 3);
        }
        e = launch_lanczos_r43(L, dt_, lanczos_exact_, th);
        if (e == hipSuccess) e = launch_lanczos_r43_edges(L, dt_, lanczos_exact_); // border columns
        break;
    }
    case Variant::LanczosPqRegWin: {
        uint32_t th = rows_per_wave_;
        if (th == 0) { // as at x4/3: 12 - 36 rows per wave on a batch, equal row blocks
            const uint32_t cols = lanczos_pq_strip_cols(pq_p_, pq_q_);
            const uint64_t rows_total = (uint64_t)ih_ * ((iw_ + cols - 1) / cols) * n_frames;
            uint64_t t = rows_total / 12288;
            t = t < 12 ? 12 : (t > 36 ? 36 : t);
            const uint64_t blocks = (ih_ + t - 1) / t;
            th = (uint32_t)(((ih_ + blocks - 1) / blocks + pq_q_ - 1) / pq_q_ * pq_q_);
        }
        if (lanczos_exact_ && pq_exact_fallback_) // (see choose_resize_variant)
            e = launch_resize_win(L, dt_, true, resize_ncols_max_, resize_union_taps_, win_outputs_per_lane_);
        else
            e = launch_lanczos_pq(L, dt_, lanczos_exact_, pq_p_, pq_q_, th, pq_narrow_ && pq_narrow_allowed_);
        break;
    }
    case Variant::LanczosR32RegWin: {
        uint32_t th = rows_per_wave_;
        if (th == 0) {
            // as at x2: enough waves to fill the chip (3 per SIMD) a few times over, tall enough to amortise the 8 halo rows; the
            // kernel's time is flat from 24 to 96 rows per wave on a batch, so the row blocks are made equal rather than tall
            // (a last block of a few rows costs more than anything else here: profiles/r03_lanczos_r32_rework_ab.txt)
            const uint64_t rows_total = (uint64_t)ih_ * ((iw_ + 239) / 240) * n_frames;
            uint64_t t = rows_total / 12288;
            t = t < 12 ? 12 : (t > 48 ? 48 : t);
            const uint64_t blocks = (ih_ + t - 1) / t;
            th = (uint32_t)((ih_ + blocks - 1) / blocks + 1) & ~1u;
        }
        e = launch_lanczos_r32(L, dt_, lanczos_exact_, th);
        if (e == hipSuccess) e = launch_lanczos_r32_edges(L, dt_, lanczos_exact_); // border columns
        break;
    }
    case Variant::LanczosXsRegWin:
    {
        uint32_t th = rows_per_wave_;
        if (th == 0) { // short, equal row blocks: 12 - 16 rows per wave measured best at x3 and x4 on batches (24 - 36: + 4 %)
            const uint64_t rows_total = (uint64_t)ih_ * ((iw_ + kLanczosX2StripCols - 1) / kLanczosX2StripCols) * n_frames;
            uint64_t t = rows_total / 16384;
            t = t < 8 ? 8 : (t > 16 ? 16 : t);
            const uint64_t blocks = (ih_ + t - 1) / t;
            th = (uint32_t)((ih_ + blocks - 1) / blocks);
        }
        e = launch_lanczos_xs(L, dt_, lanczos_exact_, xs_factor_, th);
        if (e == hipSuccess) e = launch_lanczos_xs_edges(L, dt_, lanczos_exact_, xs_factor_); // border columns
        break;
    }
    case Variant::LanczosX2RegWin: {
        // First / last 8 output columns: renormalised edge weights, row-per-lane kernel.  The main kernel leaves those columns
        // alone, so the pass depends only on what the caller's stream held before this call.  For a batch it is forked onto the
        // edge stream, launched FIRST (its single-wave blocks are on the GPU when the main kernel's blocks arrive and drain
        // beside them; launched second they would wait for the main grid to run dry) and joins the caller's stream behind the
        // main kernel.
        {
            const hipStream_t es = beside ? s_edge_ : stream;
            UpscaleLaunch E = L;
            E.stream = es;
            if (unit) { // the real frames' edge columns come from A alone, the in-between frames' from the blended pair
                UpscaleLaunch R = E;
                R.in_b = nullptr;
                e = launch_lanczos_x2_edges(R, dt_, lanczos_exact_);
                if (e != hipSuccess) return fail_hip(e, "kernel launch");
                E.out = unit->out_mid;
            }
            e = launch_lanczos_x2_edges(E, dt_, lanczos_exact_);
            if (e != hipSuccess) return fail_hip(e, "kernel launch");
            if (beside) NUS_HIP(hipEventRecord(ev_join_, s_edge_));
        }
        const uint32_t th = lanczos_x2_rows_per_wave(n_frames, unit != nullptr);
        if (unit) {
            UnitOutputs U;
            U.out_mid = unit->out_mid;
            U.mid = unit->mid;
            U.order = unit_order_;
            e = launch_lanczos_x2_unit(L, dt_, lanczos_exact_, th, U);
        } else {
            e = launch_lanczos_x2(L, dt_, lanczos_exact_, th);
        }
        break;
    }
    }
    if (e != hipSuccess) {
        if (beside) (void)hipStreamSynchronize(s_edge_); // nothing of a failed call stays in flight on the side stream
        return fail_hip(e, "kernel launch");
    }
    if (beside) NUS_HIP(hipStreamWaitEvent(stream, ev_join_, 0)); // the frames are complete when both kernels are
    // the bracket covers every launch that writes bytes of these frames (main kernel + edge columns)
    if (ev_end) NUS_HIP(hipEventRecord(ev_end, stream));
    return kOk;
}

int HipUpscaler::set_profiling(bool on)
{
    std::lock_guard<std::mutex> lk(mu_);
    profiling_ = on;
    return kOk;
}

int HipUpscaler::profile_collect(uint64_t *launches, double *total_ms)
{
    std::lock_guard<std::mutex> lk(mu_);
    double sum = 0.0;
    if (prof_used_) NUS_HIP(hipSetDevice(device_));
    for (size_t i = 0; i + 1 < prof_used_; i += 2) {
        NUS_HIP(hipEventSynchronize(prof_events_[i + 1]));
        float ms = 0.0f;
        NUS_HIP(hipEventElapsedTime(&ms, prof_events_[i], prof_events_[i + 1]));
        sum += ms;
    }
    if (launches) *launches = prof_used_ / 2;
    if (total_ms) *total_ms = sum;
    prof_used_ = 0;
    return kOk;
}

int HipUpscaler::upscale_device(const void *d_in, void *d_out, uint32_t n_frames, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) return fail(kNotInitialized, "Upscaler not initialized. Call initialize() first.");
    if (!d_in || !d_out) return fail(kInvalidArgument, "upscale_device: null device pointer");
    if (n_frames == 0) return kOk;
    if ((reinterpret_cast<uintptr_t>(d_in) % 16) || (reinterpret_cast<uintptr_t>(d_out) % 16))
        return fail(kInvalidArgument, "upscale_device: device pointers must be 16-byte aligned");
    if (n_frames > 1 && (((size_t)iw_ * ih_ * 4) % 16 || ((size_t)ow_ * oh_ * 4) % 16))
        return fail(kInvalidArgument, "upscale_device: batched frames need frame sizes that are multiples of 16 bytes");
    NUS_HIP(hipSetDevice(device_));
    return enqueue(static_cast<const uint8_t *>(d_in), static_cast<uint8_t *>(d_out), n_frames, stream);
}

int HipUpscaler::upscale_blend_device(const void *d_a, size_t a_stride, const void *d_b, size_t b_stride, float t,
                                      void *d_out, uint32_t n_frames, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) return fail(kNotInitialized, "Upscaler not initialized. Call initialize() first.");
    if (variant_ != Variant::LanczosX2RegWin)
        return fail(kUnsupported, "upscale_blend_device: only the exact-x2 resize kernels fuse the blend; "
                                  "run interpolate + upscale separately for this configuration");
    if (!d_a || !d_b || !d_out) return fail(kInvalidArgument, "upscale_blend_device: null device pointer");
    if (n_frames == 0) return kOk;
    if ((reinterpret_cast<uintptr_t>(d_a) % 16) || (reinterpret_cast<uintptr_t>(d_b) % 16) ||
        (reinterpret_cast<uintptr_t>(d_out) % 16) || (a_stride % 16) || (b_stride % 16) ||
        (n_frames > 1 && ((size_t)ow_ * oh_ * 4) % 16))
        return fail(kInvalidArgument, "upscale_blend_device: pointers and strides must be 16-byte aligned");
    NUS_HIP(hipSetDevice(device_));
    BlendSrc bs;
    bs.b = static_cast<const uint8_t *>(d_b);
    bs.a_stride = a_stride;
    bs.b_stride = b_stride;
    bs.t = t;
    return enqueue(static_cast<const uint8_t *>(d_a), static_cast<uint8_t *>(d_out), n_frames, stream, &bs);
}

int HipUpscaler::upscale_unit_device(const void *d_a, size_t a_stride, const void *d_b, size_t b_stride, float t, void *d_mid,
                                     void *d_out_real, void *d_out_mid, uint32_t n_units, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) return fail(kNotInitialized, "Upscaler not initialized. Call initialize() first.");
    if (variant_ != Variant::LanczosX2RegWin)
        return fail(kUnsupported, "upscale_unit_device: only the exact-x2 resize kernels run the whole step in one launch; "
                                  "run interpolate + upscale separately for this configuration");
    if (!d_a || !d_b || !d_out_real || !d_out_mid) return fail(kInvalidArgument, "upscale_unit_device: null device pointer");
    if (!(t >= 0.0f && t <= 1.0f)) return fail(kInvalidArgument, "upscale_unit_device: t must be in [0, 1]");
    if (n_units == 0) return kOk;
    auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) % 16) != 0; };
    if (misaligned(d_a) || misaligned(d_b) || misaligned(d_mid) || misaligned(d_out_real) || misaligned(d_out_mid) ||
        (a_stride % 16) || (b_stride % 16) ||
        (n_units > 1 && (((size_t)ow_ * oh_ * 4) % 16 || ((size_t)iw_ * ih_ * 4) % 16)))
        return fail(kInvalidArgument, "upscale_unit_device: pointers, strides and frame sizes must be multiples of 16 bytes");
    const size_t in_bytes = (size_t)iw_ * ih_ * 4;
    if ((a_stride && a_stride < in_bytes) || (b_stride && b_stride < in_bytes))
        return fail(kInvalidArgument, "upscale_unit_device: frame stride smaller than a frame");
    NUS_HIP(hipSetDevice(device_));
    BlendSrc bs;
    bs.b = static_cast<const uint8_t *>(d_b);
    bs.a_stride = a_stride ? a_stride : in_bytes;
    bs.b_stride = b_stride ? b_stride : in_bytes;
    bs.t = t;
    UnitDst ud;
    ud.out_mid = static_cast<uint8_t *>(d_out_mid);
    ud.mid = static_cast<uint8_t *>(d_mid);
    return enqueue(static_cast<const uint8_t *>(d_a), static_cast<uint8_t *>(d_out_real), n_units, stream, &bs, &ud);
}

int HipUpscaler::upscale(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap)
{
    const uint8_t *ins[1] = {in};
    const size_t lens[1] = {in_len};
    uint8_t *outs[1] = {out};
    return upscale_batch(ins, lens, 1, outs, out_cap);
}

int HipUpscaler::ensure_streams()
{
    if (s_in_ && s_k_ && s_out_) return kOk;
    if (!s_in_) NUS_HIP(hipStreamCreateWithFlags(&s_in_, hipStreamNonBlocking));
    if (!s_k_) NUS_HIP(hipStreamCreateWithFlags(&s_k_, hipStreamNonBlocking));
    if (!s_out_) NUS_HIP(hipStreamCreateWithFlags(&s_out_, hipStreamNonBlocking));
    return kOk;
}

// Events, device frames and pinned staging of one pipeline slot: all or nothing.
int HipUpscaler::ensure_slot(Slot &S, size_t in_bytes, size_t out_bytes)
{
    if (S.d_in) return kOk;
    auto make = [&]() -> int {
        NUS_HIP(hipEventCreate(&S.k_begin));
        NUS_HIP(hipEventCreate(&S.k_end));
        NUS_HIP(hipEventCreateWithFlags(&S.in_done, hipEventDisableTiming));
        NUS_HIP(hipEventCreateWithFlags(&S.out_done, hipEventDisableTiming));
        for (hipEvent_t &ev : S.chunk_done) NUS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (hipEvent_t &ev : S.band_in) NUS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (hipEvent_t &ev : S.band_k) NUS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        NUS_HIP(pinned_alloc(reinterpret_cast<void **>(&S.h_in), in_bytes));
        NUS_HIP(pinned_alloc(reinterpret_cast<void **>(&S.h_out), out_bytes));
        NUS_HIP(hipMalloc(reinterpret_cast<void **>(&S.d_out), out_bytes));
        NUS_HIP(hipMalloc(reinterpret_cast<void **>(&S.d_in), in_bytes)); // last: d_in set = the slot is complete
        return kOk;
    };
    const int rc = make();
    if (rc != kOk) release_slot(S); // a half-built slot must not look usable to the next call
    return rc;
}

void HipUpscaler::release_slot(Slot &s)
{
    for (hipStream_t st : {s_in_, s_k_, s_out_})
        if (st) (void)hipStreamSynchronize(st);
    parallel_copy_wait(s.populate); // the pool holds a pointer to this ticket for every request still queued under it
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    pinned_free(s.h_in);
    pinned_free(s.h_out);
    for (hipEvent_t ev : {s.k_begin, s.k_end, s.in_done, s.out_done})
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s.chunk_done)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s.band_in)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s.band_k)
        if (ev) (void)hipEventDestroy(ev);
    s = Slot();
}

// One frame into a free slot: stage (unless the caller's buffer is pinned) -> H2D -> kernel -> D2H, all asynchronous, each on
// its own stream.  Pageable outputs come back in kOutChunks pieces: while piece k is copied out of the pinned buffer (by
// the copy pool's workers), piece k+1 is in flight.  *direct: the D2H goes straight into the caller's (pinned) buffer.
void HipUpscaler::plan_chunks(Slot &S, size_t out_bytes, bool alone) const
{
    auto page = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
    int n = 0;
    size_t end[kOutChunks];
    if (alone && (single_out_plan_ == 1 || single_out_plan_ == 2)) {
        // halves: the copy-out of a piece has the whole DMA of the next, smaller one to finish in; what is left after the last
        // DMA is the copy of an eighth (a sixteenth) of the frame
        const int parts = single_out_plan_ == 1 ? 4 : 5;
        size_t off = 0;
        for (int k = 0; k < parts - 1; ++k) {
            off += page(out_bytes >> (k + 1)); // 1/2, 1/4, ... ; the last piece is what remains
            end[n++] = off;
        }
        end[n++] = out_bytes;
    } else {
        int parts = alone ? (single_out_plan_ == 3 ? 4 : kOutChunks) : batch_out_chunks_;
        parts = parts < 1 ? 1 : (parts > kOutChunks ? kOutChunks : parts);
        const size_t chunk = page((out_bytes + parts - 1) / parts);
        for (size_t off = chunk; off < out_bytes; off += chunk) end[n++] = off;
        end[n++] = out_bytes;
    }
    // (tiny frames: pieces that would start at or past the end are dropped)
    S.nchunks = 0;
    size_t prev = 0;
    for (int k = 0; k < n; ++k) {
        const size_t e = end[k] < out_bytes ? end[k] : out_bytes;
        if (e > prev || (k == n - 1 && S.nchunks == 0)) S.chunk_end[S.nchunks++] = e, prev = e;
    }
}

// The same for a frame that is alone in the pipeline, band by band (see kBands): band b is rows [k0, k1) of the input, a
// multiple of `align` rows; its upload carries the tap rows the kernel reads below k1 (kBandHalo, and the band after it starts
// that much later), its download is output rows [2 k0, 2 k1) in two pieces.
int HipUpscaler::submit_frame_banded(Slot &S, const uint8_t *in, uint8_t *out, bool *direct, uint32_t align, int populate)
{
    constexpr uint32_t kBandHalo = 4; // >= 3 tap rows below a row (Lanczos-3 at x2), 1 for bilinear
    const size_t in_row = (size_t)iw_ * 4, out_row = (size_t)ow_ * 4, out_bytes = out_row * oh_;
    const uint32_t band_rows = ((ih_ + kBands - 1) / kBands + align - 1) / align * align;
    const bool in_pinned = is_pinned_host(in);
    *direct = is_pinned_host(out);
    if (!*direct && (populate > 0 || (populate < 0 && parallel_populate_prepare(out, out_bytes))))
        parallel_populate_async(out, out_bytes, S.populate); // see submit_frame
    S.nchunks = 0;
    int b = 0;
    for (uint32_t k0 = 0; k0 < ih_; k0 += band_rows, ++b) {
        const uint32_t k1 = k0 + band_rows < ih_ ? k0 + band_rows : ih_;
        const uint32_t h0 = k0 == 0 ? 0 : (k0 + kBandHalo < ih_ ? k0 + kBandHalo : ih_);
        const uint32_t h1 = k1 == ih_ ? ih_ : (k1 + kBandHalo < ih_ ? k1 + kBandHalo : ih_);
        if (h1 > h0) {
            const size_t off = (size_t)h0 * in_row, len = (size_t)(h1 - h0) * in_row;
            if (!in_pinned) parallel_copy(S.h_in + off, in + off, len);
            NUS_HIP(hipMemcpyAsync(S.d_in + off, (in_pinned ? in : S.h_in) + off, len, hipMemcpyHostToDevice, s_in_));
        }
        NUS_HIP(hipEventRecord(S.band_in[b], s_in_));
        NUS_HIP(hipStreamWaitEvent(s_k_, S.band_in[b], 0));
        if (b == 0) NUS_HIP(hipEventRecord(S.k_begin, s_k_));
        const int rc = enqueue(S.d_in, S.d_out, 1, s_k_, nullptr, nullptr, k0, k1 - k0);
        if (rc != kOk) return rc;
        if (k1 == ih_) NUS_HIP(hipEventRecord(S.k_end, s_k_));
        NUS_HIP(hipEventRecord(S.band_k[b], s_k_));
        NUS_HIP(hipStreamWaitEvent(s_out_, S.band_k[b], 0));
        const size_t o0 = (size_t)2 * k0 * out_row, o1 = (size_t)2 * k1 * out_row;
        if (*direct) {
            NUS_HIP(hipMemcpyAsync(out + o0, S.d_out + o0, o1 - o0, hipMemcpyDeviceToHost, s_out_));
        } else {
            const size_t mid = (o0 + (o1 - o0) / 2 + 4095) & ~(size_t)4095;
            const size_t ends[2] = {mid < o1 ? mid : o1, o1};
            size_t off = o0;
            for (size_t e : ends) {
                if (e <= off) continue;
                NUS_HIP(hipMemcpyAsync(S.h_out + off, S.d_out + off, e - off, hipMemcpyDeviceToHost, s_out_));
                NUS_HIP(hipEventRecord(S.chunk_done[S.nchunks], s_out_));
                S.chunk_end[S.nchunks++] = e;
                off = e;
            }
        }
    }
    NUS_HIP(hipEventRecord(S.in_done, s_in_));
    NUS_HIP(hipEventRecord(S.out_done, s_out_));
    S.used = true;
    return kOk;
}

int HipUpscaler::submit_frame(Slot &S, const uint8_t *in, uint8_t *out, bool *direct, bool alone, int populate)
{
    const int rc = submit_frame_inner(S, in, out, direct, alone, populate);
    // a frame that was not submitted is never retired: its populate requests (they point into `out`) end here
    if (rc != kOk) parallel_copy_wait(S.populate);
    return rc;
}

int HipUpscaler::submit_frame_inner(Slot &S, const uint8_t *in, uint8_t *out, bool *direct, bool alone, int populate)
{
    const size_t in_bytes = (size_t)iw_ * ih_ * 4, out_bytes = (size_t)ow_ * oh_ * 4;
    if (alone && single_bands_ && !profiling_ && out_bytes >= ((size_t)8 << 20)) {
        const uint32_t align = band_alignment(1);
        if (align && (uint64_t)align * kBands * 2 <= ih_ && ow_ == 2 * iw_ && oh_ == 2 * ih_)
            return submit_frame_banded(S, in, out, direct, align, populate);
    }
    // First of all, and in the pool's low-priority queue (behind every copy): while the frame is staged, uploaded, computed and
    // on the wire back, idle workers make the caller's output pages present -- a result buffer fresh from the allocator (the Vec /
    // PyBytes `upscale` returns) is 8 100 first-touch faults, or 16 huge-page faults, that would otherwise be taken inside the
    // copy-out; pages that are resident already cost a page-table walk.
    // `populate`: 1 = the caller has prepared the buffer (upscale_batch does that for all its outputs before the first request
    // is queued: parallel_populate_prepare), 0 = it found it resident, -1 = find out here.
    *direct = is_pinned_host(out);
    if (!*direct && (populate > 0 || (populate < 0 && parallel_populate_prepare(out, out_bytes))))
        parallel_populate_async(out, out_bytes, S.populate);
    const uint8_t *src = in;
    if (!is_pinned_host(src)) {
        parallel_copy(S.h_in, src, in_bytes);
        src = S.h_in;
    }
    NUS_HIP(hipMemcpyAsync(S.d_in, src, in_bytes, hipMemcpyHostToDevice, s_in_));
    NUS_HIP(hipEventRecord(S.in_done, s_in_));
    NUS_HIP(hipStreamWaitEvent(s_k_, S.in_done, 0));
    NUS_HIP(hipEventRecord(S.k_begin, s_k_));
    int rc = enqueue(S.d_in, S.d_out, 1, s_k_);
    if (rc != kOk) return rc;
    NUS_HIP(hipEventRecord(S.k_end, s_k_));
    NUS_HIP(hipStreamWaitEvent(s_out_, S.k_end, 0));
    if (*direct) {
        NUS_HIP(hipMemcpyAsync(out, S.d_out, out_bytes, hipMemcpyDeviceToHost, s_out_));
    } else {
        plan_chunks(S, out_bytes, alone);
        size_t off = 0;
        for (int k = 0; k < S.nchunks; ++k) {
            NUS_HIP(hipMemcpyAsync(S.h_out + off, S.d_out + off, S.chunk_end[k] - off, hipMemcpyDeviceToHost, s_out_));
            NUS_HIP(hipEventRecord(S.chunk_done[k], s_out_));
            off = S.chunk_end[k];
        }
    }
    NUS_HIP(hipEventRecord(S.out_done, s_out_));
    S.used = true;
    return kOk;
}

// Waits for the slot's frame and hands its bytes to `out`.  Runs on the retiring thread: touches no member but the slot, and
// reports an error's text through *err instead of error_.
int HipUpscaler::retire_frame(Slot &S, uint8_t *out, bool direct, std::string *err) const
{
    auto hip_failed = [&](hipError_t e, const char *what) {
        (void)hipGetLastError();
        if (err) *err = fmt("HIP error in %s: %s", what, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? kOutOfMemory : kHipError;
    };
    // (test hook, option "inject_retire_error": this frame's wait reports a failure, as a lost device would)
    const bool injected = inject_retire_.load() > 0 && inject_retire_.fetch_sub(1) == 1;
    if (direct) {
        const hipError_t e = injected ? hipErrorUnknown : hipEventSynchronize(S.out_done);
        return e == hipSuccess ? kOk : hip_failed(e, "hipEventSynchronize(out_done)");
    }
    struct PopulateDone { // on every way out: queued populate requests point into `out`
        CopyTicket &t;
        ~PopulateDone() { parallel_copy_wait(t); }
    } populate_done{S.populate};
    CopyTicket ticket; // the frame's pieces: queued as they land, all copied when the wait returns
    int rc = kOk;
    size_t off = 0;
    for (int k = 0; k < S.nchunks; ++k) {
        const hipError_t e = injected ? hipErrorUnknown : hipEventSynchronize(S.chunk_done[k]);
        if (e != hipSuccess) {
            rc = hip_failed(e, "hipEventSynchronize(chunk_done)");
            break;
        }
        parallel_copy_async(out + off, S.h_out + off, S.chunk_end[k] - off, ticket);
        off = S.chunk_end[k];
    }
    parallel_copy_wait(ticket); // also on the error path: queued pieces point into buffers that must outlive them
    return rc;
}

// ---- persistent ring: frames go in one at a time and come out in order, up to kSlots in flight ---------------------------
// (the shape of a capture loop: Nu_scale/src/capture/frame_buffer.rs:11-50 feeding upscale(); rayon's fan-out over a batch is
// upscale_batch above).  The retiring thread lives from stream_open to stream_close and touches only the ring's own state.

int HipUpscaler::stream_open()
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) return fail(kNotInitialized, "Upscaler not initialized. Call initialize() first.");
    if (ring_.open) return fail(kInvalidArgument, "stream_open: a stream is already open on this upscaler");
    NUS_HIP(hipSetDevice(device_));
    int rc = ensure_streams();
    for (int s = 0; s < kSlots && rc == kOk; ++s) rc = ensure_slot(slots_[s], (size_t)iw_ * ih_ * 4, (size_t)ow_ * oh_ * 4);
    if (rc != kOk) return rc;
    {
        std::lock_guard<std::mutex> rl(ring_.m);
        ring_.submitted = ring_.retired = 0;
        ring_.stop = false;
        ring_.status = kOk;
        ring_.error.clear();
    }
    ring_.thread = std::thread([this] {
        (void)hipSetDevice(device_);
        for (uint64_t i = 0;; ++i) {
            Ring::Item item;
            {
                std::unique_lock<std::mutex> rl(ring_.m);
                ring_.cv.wait(rl, [&] { return ring_.submitted > i || ring_.stop; });
                if (ring_.submitted <= i) return; // stop, and nothing left in flight
                item = ring_.items[i % kSlots];
            }
            std::string err;
            int rc;
            try {
                rc = retire_frame(slots_[i % kSlots], item.out, item.direct, &err);
            } catch (...) {
                rc = kOutOfMemory;
            }
            std::lock_guard<std::mutex> rl(ring_.m);
            if (rc != kOk && ring_.status == kOk) {
                ring_.status = rc;
                ring_.error.swap(err);
            }
            ring_.retired = i + 1;
            ring_.cv.notify_all();
        }
    });
    ring_.open = true;
    return kOk;
}

int HipUpscaler::stream_submit(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, uint64_t *ticket)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!ring_.open) return fail(kInvalidArgument, "stream_submit: no stream is open (call stream_open first)");
    if (!in || !out) return fail(kInvalidArgument, "stream_submit: null frame pointer");
    const size_t in_bytes = (size_t)iw_ * ih_ * 4, out_bytes = (size_t)ow_ * oh_ * 4;
    if (in_len != in_bytes)
        return fail(kSizeMismatch, fmt("Input data size (%zu) does not match expected input buffer size (%zu for %ux%u)", in_len,
                                       in_bytes, iw_, ih_));
    if (out_cap < out_bytes)
        return fail(kInvalidArgument, fmt("Output capacity (%zu) is smaller than the output frame (%zu for %ux%u)", out_cap,
                                          out_bytes, ow_, oh_));
    uint64_t i;
    {
        std::unique_lock<std::mutex> rl(ring_.m);
        if (ring_.status != kOk) return fail(ring_.status, ring_.error);
        i = ring_.submitted;
        ring_.cv.wait(rl, [&] { return ring_.retired + (uint64_t)kSlots > i || ring_.status != kOk; }); // a free slot
        if (ring_.status != kOk) return fail(ring_.status, ring_.error);
    }
    NUS_HIP(hipSetDevice(device_));
    bool direct = false;
    const int rc = submit_frame(slots_[i % kSlots], in, out, &direct, false);
    if (rc != kOk) { // nothing was queued for the retiring thread: the ring stays consistent.  Copies of this frame may be
                     // queued on the caller's (pinned) buffers already: nothing of it may be in flight when the call returns
        for (hipStream_t st : {s_in_, s_k_, s_out_})
            if (st) (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        return rc;
    }
    {
        std::lock_guard<std::mutex> rl(ring_.m);
        ring_.items[i % kSlots] = Ring::Item{out, direct};
        ring_.submitted = i + 1;
    }
    ring_.cv.notify_all();
    if (ticket) *ticket = i;
    return kOk;
}

int HipUpscaler::stream_wait(uint64_t ticket)
{
    // (not under mu_: a second thread may wait for results while the first one is blocked in stream_submit)
    std::unique_lock<std::mutex> rl(ring_.m);
    if (ticket >= ring_.submitted) {
        set_thread_error("stream_wait: no such frame has been submitted");
        return kInvalidArgument;
    }
    // Always until THIS frame has been retired, sticky error or not: the retiring thread keeps retiring after a failure (it
    // always advances `retired`), and until it has passed `ticket` a D2H or a pool worker may still be writing `out`.
    ring_.cv.wait(rl, [&] { return ring_.retired > ticket; });
    if (ring_.status == kOk) return kOk;
    set_thread_error(ring_.error);
    return ring_.status;
}

int HipUpscaler::stream_close()
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!ring_.open) return kOk;
    {
        std::lock_guard<std::mutex> rl(ring_.m);
        ring_.stop = true; // the thread first retires everything that was submitted
    }
    ring_.cv.notify_all();
    if (ring_.thread.joinable()) ring_.thread.join();
    ring_.open = false;
    for (hipStream_t st : {s_in_, s_k_, s_out_})
        if (st) (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    std::lock_guard<std::mutex> rl(ring_.m);
    return ring_.status == kOk ? kOk : fail(ring_.status, ring_.error);
}

int HipUpscaler::upscale_batch(const uint8_t *const *ins, const size_t *in_lens, size_t n, uint8_t *const *outs,
                               size_t out_cap_each)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) return fail(kNotInitialized, "Upscaler not initialized. Call initialize() first.");
    if (ring_.open) return fail(kInvalidArgument, "upscale: a stream is open on this upscaler (its slots are in use); close it first");
    if (n == 0) return kOk;
    if (!ins || !in_lens || !outs) return fail(kInvalidArgument, "upscale: null argument");
    const size_t in_bytes = (size_t)iw_ * ih_ * 4, out_bytes = (size_t)ow_ * oh_ * 4;
    for (size_t i = 0; i < n; ++i) {
        if (!ins[i] || !outs[i]) return fail(kInvalidArgument, "upscale: null frame pointer");
        if (in_lens[i] != in_bytes)
            return fail(kSizeMismatch, fmt("Input data size (%zu) does not match expected input buffer size (%zu for %ux%u)",
                                           in_lens[i], in_bytes, iw_, ih_));
    }
    if (out_cap_each < out_bytes)
        return fail(kInvalidArgument, fmt("Output capacity (%zu) is smaller than the output frame (%zu for %ux%u)",
                                          out_cap_each, out_bytes, ow_, oh_));
    NUS_HIP(hipSetDevice(device_));
    const int nslots = n < (size_t)kSlots ? (int)n : kSlots;
    {
        int rc = ensure_streams();
        if (rc != kOk) return rc;
    }
    for (int s = 0; s < nslots; ++s) {
        int rc = ensure_slot(slots_[s], in_bytes, out_bytes);
        if (rc != kOk) return rc;
    }
    // whatever way this call ends, nothing may still be reading the caller's input or writing its
    // (pinned) output buffers afterwards
    struct Drain {
        hipStream_t st[3];
        Slot *slots;
        int nslots;
        bool armed = true;
        ~Drain()
        {
            if (!armed) return;
            for (hipStream_t s : st)
                if (s) (void)hipStreamSynchronize(s);
            (void)hipGetLastError();
            // and no populate request of this call may stay queued: they point into outs[] (the caller frees those next) and
            // into the slots' tickets (a request left behind would later decrement a count inside a freed HipUpscaler)
            for (int s = 0; s < nslots; ++s) parallel_copy_wait(slots[s].populate);
        }
    } drain{{s_in_, s_k_, s_out_}, slots_, nslots};
    // Frame i uses slot i % nslots: stage -> H2D (copy-in stream) -> kernel (compute stream) -> D2H (copy-out stream),
    // chained by the slot's events (submit_frame / retire_frame).  A slot is submitted to again only after its previous frame
    // has been retired (so its D2H, hence its kernel and its H2D, are complete): no stream needs to wait for an earlier frame.
    std::vector<char> direct_out(n, 0);
    // Output buffers fresh from the allocator (the Vecs / PyBytes of the trait's `upscale_batch`) are found and given their
    // huge-page hint here, before the first populate request exists: the hint takes the address space's lock exclusively.
    std::vector<signed char> populate(n, 0);
    for (size_t i = 0; i < n; ++i)
        populate[i] = !is_pinned_host(outs[i]) && parallel_populate_prepare(outs[i], out_bytes) ? 1 : 0;
    auto submit = [&](size_t i) -> int {
        bool direct = false;
        const int rc = submit_frame(slots_[i % nslots], ins[i], outs[i], &direct, n == 1, populate[i]);
        direct_out[i] = direct ? 1 : 0;
        return rc;
    };
    // (also runs on the retiring thread: the error text stays in *err there; fail() -- error_ and the thread-local text -- is
    // for the calling thread only, after the join)
    auto retire = [&](size_t i, std::string *err) -> int {
        return retire_frame(slots_[i % nslots], outs[i], direct_out[i] != 0, err);
    };
    int status = kOk;
    if (n == 1) { // one frame (trait Upscaler::upscale): nothing to overlap with, no second thread
        status = submit(0);
        if (status == kOk) {
            std::string err;
            status = retire(0, &err);
            if (status != kOk) status = fail(status, err);
        }
    } else {
        // The calling thread stages and submits frame i as soon as slot i % nslots is free again; a second thread retires the
        // frames in order.  The copy-out of frame i (the pool's workers) then runs beside the staging copy, the H2D and the
        // kernel of frames i+1 .. i+nslots-1, and the D2H engine goes from one frame's bytes straight to the next one's.
        std::mutex m;
        std::condition_variable cv;
        size_t submitted = 0, retired = 0; // frames handed to the GPU / copied out to the caller
        bool stop = false;                 // the submitter gave up: retire what was submitted, then leave
        int retire_status = kOk;
        std::string retire_error;
        auto retire_all = [&] {
            (void)hipSetDevice(device_);
            for (size_t i = 0; i < n; ++i) {
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return submitted > i || stop; });
                    if (submitted <= i) return;
                }
                int rc;
                std::string err;
                try {
                    rc = retire(i, &err);
                } catch (...) { // (an allocation failing while an error text is built: no exception may leave a thread)
                    rc = kOutOfMemory;
                }
                std::lock_guard<std::mutex> lk(m);
                if (rc != kOk && retire_status == kOk) {
                    retire_status = rc;
                    retire_error.swap(err);
                }
                retired = i + 1;
                cv.notify_all();
                // (no early return on a failed frame: every frame that was SUBMITTED is retired -- retire_frame is what waits for
                // the populate requests and copy pieces that point into the caller's outs[i] and into the slot's ticket; the
                // submitter stops submitting as soon as it sees retire_status)
            }
        };
        // the retiring thread is told to stop and joined on EVERY way out of this scope (an exception on the submitting
        // side included): a joinable std::thread must not reach its destructor
        struct Retirer {
            std::thread t;
            std::mutex &m;
            std::condition_variable &cv;
            bool &stop;
            ~Retirer()
            {
                {
                    std::lock_guard<std::mutex> lk(m);
                    stop = true;
                }
                cv.notify_all();
                if (t.joinable()) t.join();
            }
        } retirer{std::thread(retire_all), m, cv, stop};
        for (size_t i = 0; i < n && status == kOk; ++i) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return retired + (size_t)nslots > i || retire_status != kOk; });
                if (retire_status != kOk) break;
            }
            status = submit(i);
            if (status != kOk) break;
            std::lock_guard<std::mutex> lk(m);
            submitted = i + 1;
            cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        retirer.t.join();
        if (status == kOk && retire_status != kOk) status = fail(retire_status, retire_error);
    }
    if (status != kOk) return status; // ~Drain waits for whatever is still in flight
    drain.armed = false;             // every frame was retired above
    float ms = 0.0f;
    Slot &last = slots_[(n - 1) % nslots];
    if (hipEventElapsedTime(&ms, last.k_begin, last.k_end) == hipSuccess) {
        have_ms_ = true;
        last_ms_ = ms;
    } else {
        (void)hipGetLastError();
    }
    return kOk;
}

bool HipUpscaler::last_gpu_ms(double *ms) const
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!have_ms_) return false;
    if (ms) *ms = last_ms_;
    return true;
}

int64_t HipUpscaler::export_tables(void *buf, size_t cap) const
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) {
        set_thread_error("export_tables: upscaler not initialized");
        return kNotInitialized;
    }
    const std::vector<uint8_t> blob = serialize_tables(tx_, ty_);
    if (buf) {
        if (cap < blob.size()) {
            set_thread_error("export_tables: buffer too small");
            return kInvalidArgument;
        }
        memcpy(buf, blob.data(), blob.size());
    }
    return (int64_t)blob.size();
}

int HipUpscaler::import_tables(const void *buf, size_t len)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!initialized_) return fail(kNotInitialized, "import_tables: upscaler not initialized");
    if (!buf) return fail(kInvalidArgument, "import_tables: null buffer");
    AxisTables x, y;
    std::string err;
    if (!deserialize_tables(static_cast<const uint8_t *>(buf), len, x, y, err)) return fail(kInvalidArgument, err);
    if (x.in_n != iw_ || x.out_n != ow_ || y.in_n != ih_ || y.out_n != oh_)
        return fail(kInvalidArgument, "import_tables: tables were built for different dimensions");
    NUS_HIP(hipSetDevice(device_));
    NUS_HIP(hipDeviceSynchronize());
    if (is_resize() && (x.filter != resize_filter() || y.filter != resize_filter()))
        return fail(kInvalidArgument, "import_tables: tables were built for a different resize filter");
    if (is_resize() && (x.lz_max_taps < 0 || y.lz_max_taps < 0))
        return fail(kUnsupported, "import_tables: Lanczos window too wide");
    // every check has passed: only now replace the tables this upscaler runs on
    tx_ = std::move(x);
    ty_ = std::move(y);
    choose_variant();
    const int rc = upload_tables();
    if (rc != kOk) initialized_ = false; // the device tables are gone: the upscaler must be initialised again
    return rc;
}

std::unique_ptr<HipUpscaler> UpscalerFactory::create_upscaler(Technology tech, Quality q)
{
    // upscale/mod.rs:99-116: Wgpu -> bilinear, everything else -> nearest.
    const Algorithm a = tech == Technology::Wgpu ? Algorithm::Bilinear : Algorithm::Nearest;
    return std::unique_ptr<HipUpscaler>(new HipUpscaler(q, a));
}

} // namespace nus

// nus_capi.cpp -- extern "C" boundary (include/nuscaler_hip.h) over the host classes.
#include "../../include/nuscaler_hip.h"

#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "nus_flow.hpp"
#include "nus_host.hpp"
#include "nus_queue.hpp"
#include "nus_ranges.hpp"
#include "nus_transfer.hpp"

struct nus_upscaler {
    nus::HipUpscaler impl;
    nus_upscaler(nus::Quality q, nus::Algorithm a) : impl(q, a) {}
};

struct nus_frame_queue {
    nus::FrameQueue impl;
    explicit nus_frame_queue(size_t cap) : impl(cap) {}
};

struct nus_flow {
    nus::HipFlowEstimator impl;
};

struct nus_interp {
    nus::HipFrameInterpolator impl;
    explicit nus_interp(int preset) : impl(preset) {}
};

namespace {
int null_handle()
{
    nus::set_thread_error("null handle");
    return NUS_ERR_INVALID_ARGUMENT;
}

// No C++ exception may cross the extern "C" boundary (the callers are C, Rust through the -sys crate, ctypes):
// every entry point that can allocate runs inside this guard.  std::bad_alloc (a vector or string that cannot grow,
// a queue frame) becomes NUS_ERR_OUT_OF_MEMORY, anything else NUS_ERR_INVALID_ARGUMENT, with the text in
// nus_last_error(); entry points that return a handle return NULL.
template <typename R>
struct GuardFail;
template <>
struct GuardFail<int> {
    static int value(int code) { return code; }
};
template <>
struct GuardFail<int64_t> {
    static int64_t value(int code) { return code; }
};
template <typename T>
struct GuardFail<T *> {
    static T *value(int) { return nullptr; }
};
template <typename R, typename F>
R guarded(const char *what, F &&f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        nus::set_thread_error("out of memory"); // short enough never to allocate itself
        return GuardFail<R>::value(NUS_ERR_OUT_OF_MEMORY);
    } catch (const std::exception &e) {
        nus::set_thread_error(std::string(what) + ": " + e.what());
        return GuardFail<R>::value(NUS_ERR_INVALID_ARGUMENT);
    } catch (...) {
        nus::set_thread_error(std::string(what) + ": unknown exception");
        return GuardFail<R>::value(NUS_ERR_INVALID_ARGUMENT);
    }
}
} // namespace

extern "C" {

int nus_abi_version(void) { return NUS_ABI_VERSION; }

int nus_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int nus_host_pin(void *buffer, size_t bytes)
{
    return guarded<int>("nus_host_pin", [&]() -> int {
        if (!buffer || bytes == 0) {
            nus::set_thread_error("nus_host_pin: null buffer");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        if (nus_device_count() <= 0) {
            nus::set_thread_error("nus_host_pin: no HIP device available");
            return NUS_ERR_NO_DEVICE;
        }
        const hipError_t e = hipHostRegister(buffer, bytes, hipHostRegisterPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            nus::set_thread_error(std::string("nus_host_pin: hipHostRegister failed: ") + hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? NUS_ERR_OUT_OF_MEMORY : NUS_ERR_HIP;
        }
        nus::range_note(nus::kRangePinned, buffer, bytes);
        return NUS_OK;
    });
}

int nus_host_unpin(void *buffer)
{
    return guarded<int>("nus_host_unpin", [&]() -> int {
        if (!buffer) {
            nus::set_thread_error("nus_host_unpin: null buffer");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        if (!nus::range_is_live(nus::kRangePinned, buffer)) {
            nus::set_thread_error("nus_host_unpin: this pointer is not the start of a buffer pinned with nus_host_pin");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        const hipError_t e = hipHostUnregister(buffer);
        if (e != hipSuccess) { // the entry stays in the record: the registration may still exist
            (void)hipGetLastError();
            nus::set_thread_error(std::string("nus_host_unpin: hipHostUnregister failed: ") + hipGetErrorString(e));
            return NUS_ERR_HIP;
        }
        nus::range_forget(nus::kRangePinned, buffer);
        return NUS_OK;
    });
}

int nus_download(void *host_dst, const void *d_src, size_t bytes, void *stream)
{
    return guarded<int>("nus_download", [&]() -> int { return nus::download(host_dst, d_src, bytes, static_cast<hipStream_t>(stream)); });
}

int nus_upload(void *d_dst, const void *host_src, size_t bytes, void *stream)
{
    return guarded<int>("nus_upload", [&]() -> int { return nus::upload(d_dst, host_src, bytes, static_cast<hipStream_t>(stream)); });
}

int nus_device_memory_info(int device, uint64_t *free_bytes, uint64_t *total_bytes)
{
    return guarded<int>("nus_device_memory_info", [&]() -> int {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
            (void)hipGetLastError();
            nus::set_thread_error("nus_device_memory_info: no such HIP device");
            return NUS_ERR_NO_DEVICE;
        }
        size_t f = 0, t = 0;
        if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) {
            (void)hipGetLastError();
            nus::set_thread_error("nus_device_memory_info: hipMemGetInfo failed");
            return NUS_ERR_HIP;
        }
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
        return NUS_OK;
    });
}

const char *nus_last_error(void) { return nus::thread_error(); }

const char *nus_status_string(int status)
{
    switch (status) {
    case NUS_OK: return "ok";
    case NUS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case NUS_ERR_NOT_INITIALIZED: return "not initialized";
    case NUS_ERR_SIZE_MISMATCH: return "size mismatch";
    case NUS_ERR_HIP: return "HIP runtime error";
    case NUS_ERR_NO_DEVICE: return "no HIP device";
    case NUS_ERR_UNSUPPORTED: return "unsupported";
    case NUS_ERR_OUT_OF_MEMORY: return "out of memory";
    default: return "unknown status";
    }
}

nus_upscaler *nus_upscaler_create(int algorithm, int quality)
{
    return guarded<nus_upscaler *>("nus_upscaler_create", [&]() -> nus_upscaler * {
        if (algorithm < NUS_ALG_NEAREST || algorithm > NUS_ALG_FSR_RCAS || quality < NUS_QUALITY_ULTRA_PERFORMANCE ||
            quality > NUS_QUALITY_NATIVE) {
            nus::set_thread_error("nus_upscaler_create: unknown algorithm or quality");
            return nullptr;
        }
        return new (std::nothrow) nus_upscaler(static_cast<nus::Quality>(quality), static_cast<nus::Algorithm>(algorithm));
    });
}

nus_upscaler *nus_upscaler_create_for_technology(int technology, int quality)
{
    return guarded<nus_upscaler *>("nus_upscaler_create_for_technology", [&]() -> nus_upscaler * {
        if (technology < NUS_TECH_NONE || technology > NUS_TECH_FALLBACK || quality < NUS_QUALITY_ULTRA_PERFORMANCE ||
            quality > NUS_QUALITY_NATIVE) {
            nus::set_thread_error("nus_upscaler_create_for_technology: unknown technology or quality");
            return nullptr;
        }
        // UpscalerFactory::create_upscaler (upscale/mod.rs:95-117)
        const int alg = technology == NUS_TECH_WGPU ? NUS_ALG_BILINEAR : NUS_ALG_NEAREST;
        return nus_upscaler_create(alg, quality);
    });
}

void nus_upscaler_destroy(nus_upscaler *h) { delete h; }

int nus_upscaler_set_device(nus_upscaler *h, int device)
{
    return guarded<int>("nus_upscaler_set_device", [&]() -> int { return h ? h->impl.set_device(device) : null_handle(); });
}
int nus_upscaler_set_bilinear_variant(nus_upscaler *h, int v)
{
    return guarded<int>("nus_upscaler_set_bilinear_variant", [&]() -> int { return h ? h->impl.set_bilinear_variant(v) : null_handle(); });
}
int nus_upscaler_set_lanczos_mode(nus_upscaler *h, int m)
{
    return guarded<int>("nus_upscaler_set_lanczos_mode", [&]() -> int { return h ? h->impl.set_lanczos_mode(m) : null_handle(); });
}
int nus_upscaler_set_option(nus_upscaler *h, const char *key, int64_t value)
{
    return guarded<int>("nus_upscaler_set_option", [&]() -> int { return h ? h->impl.set_option(key, value) : null_handle(); });
}
int nus_upscaler_get_option(nus_upscaler *h, const char *key, int64_t *value)
{
    return guarded<int>("nus_upscaler_get_option", [&]() -> int { return h ? h->impl.get_option(key, value) : null_handle(); });
}
int nus_upscaler_set_input_format(nus_upscaler *h, int format)
{
    return guarded<int>("nus_upscaler_set_input_format", [&]() -> int { return h ? h->impl.set_input_format(format) : null_handle(); });
}
int nus_upscaler_set_sharpness(nus_upscaler *h, float easu, float rcas)
{
    return guarded<int>("nus_upscaler_set_sharpness", [&]() -> int { return h ? h->impl.set_sharpness(easu, rcas) : null_handle(); });
}
int nus_upscaler_get_sharpness(const nus_upscaler *h, float *easu, float *rcas)
{
    return guarded<int>("nus_upscaler_get_sharpness", [&]() -> int {
        if (!h) return null_handle();
        if (easu) *easu = h->impl.easu_sharpness();
        if (rcas) *rcas = h->impl.rcas_sharpness();
        return NUS_OK;
    });
}

int nus_upscaler_initialize(nus_upscaler *h, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h)
{
    return guarded<int>("nus_upscaler_initialize", [&]() -> int { return h ? h->impl.initialize(in_w, in_h, out_w, out_h) : null_handle(); });
}

int nus_upscaler_upscale(nus_upscaler *h, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap)
{
    return guarded<int>("nus_upscaler_upscale", [&]() -> int { return h ? h->impl.upscale(in, in_len, out, out_cap) : null_handle(); });
}

int nus_upscaler_upscale_batch(nus_upscaler *h, const uint8_t *const *ins, const size_t *in_lens, size_t n,
                               uint8_t *const *outs, size_t out_cap_each)
{
    return guarded<int>("nus_upscaler_upscale_batch", [&]() -> int { return h ? h->impl.upscale_batch(ins, in_lens, n, outs, out_cap_each) : null_handle(); });
}

int nus_upscaler_stream_open(nus_upscaler *h)
{
    return guarded<int>("nus_upscaler_stream_open", [&]() -> int { return h ? h->impl.stream_open() : null_handle(); });
}

int nus_upscaler_stream_submit(nus_upscaler *h, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, uint64_t *ticket)
{
    return guarded<int>("nus_upscaler_stream_submit",
                        [&]() -> int { return h ? h->impl.stream_submit(in, in_len, out, out_cap, ticket) : null_handle(); });
}

int nus_upscaler_stream_wait(nus_upscaler *h, uint64_t ticket)
{
    return guarded<int>("nus_upscaler_stream_wait", [&]() -> int { return h ? h->impl.stream_wait(ticket) : null_handle(); });
}

int nus_upscaler_stream_close(nus_upscaler *h)
{
    return guarded<int>("nus_upscaler_stream_close", [&]() -> int { return h ? h->impl.stream_close() : null_handle(); });
}

int nus_upscaler_upscale_device(nus_upscaler *h, const void *d_in, void *d_out, uint32_t n_frames, void *stream)
{
    return guarded<int>("nus_upscaler_upscale_device", [&]() -> int { return h ? h->impl.upscale_device(d_in, d_out, n_frames, static_cast<hipStream_t>(stream)) : null_handle(); });
}

int nus_upscaler_upscale_blend_device(nus_upscaler *h, const void *d_a, size_t a_stride, const void *d_b, size_t b_stride,
                                      float t, void *d_out, uint32_t n_frames, void *stream)
{
    return guarded<int>("nus_upscaler_upscale_blend_device", [&]() -> int {
        return h ? h->impl.upscale_blend_device(d_a, a_stride, d_b, b_stride, t, d_out, n_frames, static_cast<hipStream_t>(stream))
                 : null_handle();
    });
}

int nus_upscaler_upscale_unit_device(nus_upscaler *h, const void *d_a, size_t a_stride, const void *d_b, size_t b_stride, float t,
                                     void *d_mid, void *d_out_real, void *d_out_mid, uint32_t n_units, void *stream)
{
    return guarded<int>("nus_upscaler_upscale_unit_device", [&]() -> int {
        return h ? h->impl.upscale_unit_device(d_a, a_stride, d_b, b_stride, t, d_mid, d_out_real, d_out_mid, n_units,
                                               static_cast<hipStream_t>(stream))
                 : null_handle();
    });
}

const char *nus_upscaler_name(const nus_upscaler *h) { return h ? h->impl.name() : ""; }
int nus_upscaler_algorithm(const nus_upscaler *h) { return h ? static_cast<int>(h->impl.algorithm()) : null_handle(); }
int nus_upscaler_quality(const nus_upscaler *h) { return h ? static_cast<int>(h->impl.quality()) : null_handle(); }

int nus_upscaler_set_quality(nus_upscaler *h, int quality)
{
    return guarded<int>("nus_upscaler_set_quality", [&]() -> int {
        if (!h) return null_handle();
        if (quality < NUS_QUALITY_ULTRA_PERFORMANCE || quality > NUS_QUALITY_NATIVE) {
            nus::set_thread_error("set_quality: unknown quality");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        return h->impl.set_quality(static_cast<nus::Quality>(quality));
    });
}

int nus_upscaler_is_initialized(const nus_upscaler *h) { return h && h->impl.initialized() ? 1 : 0; }
size_t nus_upscaler_input_size(const nus_upscaler *h) { return h ? h->impl.input_size() : 0; }
size_t nus_upscaler_output_size(const nus_upscaler *h) { return h ? h->impl.output_size() : 0; }
const char *nus_upscaler_last_error(const nus_upscaler *h) { return h ? h->impl.last_error() : "null handle"; }

int nus_upscaler_last_gpu_ms(const nus_upscaler *h, double *ms_out)
{
    return guarded<int>("nus_upscaler_last_gpu_ms", [&]() -> int {
        if (!h) return null_handle();
        return h->impl.last_gpu_ms(ms_out) ? NUS_OK : NUS_ERR_NOT_INITIALIZED;
    });
}

int nus_upscaler_set_profiling(nus_upscaler *h, int enabled)
{
    return guarded<int>("nus_upscaler_set_profiling", [&]() -> int { return h ? h->impl.set_profiling(enabled != 0) : null_handle(); });
}

int nus_upscaler_profile_collect(nus_upscaler *h, uint64_t *launches, double *total_ms)
{
    return guarded<int>("nus_upscaler_profile_collect", [&]() -> int { return h ? h->impl.profile_collect(launches, total_ms) : null_handle(); });
}

const char *nus_upscaler_kernel_variant(const nus_upscaler *h) { return h ? h->impl.kernel_variant() : ""; }

int64_t nus_upscaler_export_tables(const nus_upscaler *h, void *buf, size_t cap)
{
    return guarded<int64_t>("nus_upscaler_export_tables", [&]() -> int64_t { return h ? h->impl.export_tables(buf, cap) : null_handle(); });
}

int nus_upscaler_import_tables(nus_upscaler *h, const void *buf, size_t len)
{
    return guarded<int>("nus_upscaler_import_tables", [&]() -> int { return h ? h->impl.import_tables(buf, len) : null_handle(); });
}

int64_t nus_tables_build_blob(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h, int variant, void *buf,
                              size_t cap)
{
    return guarded<int64_t>("nus_tables_build_blob", [&]() -> int64_t { return nus_tables_build_blob_for(NUS_ALG_LANCZOS3, in_w, in_h, out_w, out_h, variant, buf, cap); });
}

int64_t nus_tables_build_blob_for(int algorithm, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                                  int variant, void *buf, size_t cap)
{
    return guarded<int64_t>("nus_tables_build_blob_for", [&]() -> int64_t {
        if (in_w == 0 || in_h == 0 || out_w == 0 || out_h == 0 || (variant != 0 && variant != 1) ||
            algorithm < NUS_ALG_NEAREST || algorithm > NUS_ALG_TRIANGLE) {
            nus::set_thread_error("nus_tables_build_blob: bad argument");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        const nus::ResizeFilter filter = algorithm == NUS_ALG_BICUBIC    ? nus::ResizeFilter::CatmullRom
                                         : algorithm == NUS_ALG_TRIANGLE ? nus::ResizeFilter::Triangle
                                                                         : nus::ResizeFilter::Lanczos3;
        nus::AxisTables x, y;
        nus::build_axis_tables(in_w, out_w, variant == 1, x, filter);
        nus::build_axis_tables(in_h, out_h, variant == 1, y, filter);
        const std::vector<uint8_t> blob = nus::serialize_tables(x, y);
        if (buf) {
            if (cap < blob.size()) {
                nus::set_thread_error("nus_tables_build_blob: buffer too small");
                return NUS_ERR_INVALID_ARGUMENT;
            }
            memcpy(buf, blob.data(), blob.size());
        }
        return (int64_t)blob.size();
    });
}

int nus_tables_validate_blob(const void *buf, size_t len, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h)
{
    return guarded<int>("nus_tables_validate_blob", [&]() -> int {
        nus::AxisTables x, y;
        std::string err;
        if (!buf || !nus::deserialize_tables(static_cast<const uint8_t *>(buf), len, x, y, err)) {
            nus::set_thread_error(buf ? err : "nus_tables_validate_blob: null buffer");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        if (x.in_n != in_w || x.out_n != out_w || y.in_n != in_h || y.out_n != out_h) {
            nus::set_thread_error("table blob was built for different dimensions");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        return NUS_OK;
    });
}

int nus_lanczos3_build_axis(uint32_t in_n, uint32_t out_n, int32_t *left, uint32_t *ntaps, float *weights)
{
    return guarded<int>("nus_lanczos3_build_axis", [&]() -> int {
        if (!left || !ntaps || !weights || in_n == 0 || out_n == 0) {
            nus::set_thread_error("nus_lanczos3_build_axis: bad argument");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        const int r = nus::build_lanczos3_axis(in_n, out_n, left, ntaps, weights);
        if (r < 0) {
            nus::set_thread_error("nus_lanczos3_build_axis: window exceeds NUS_RESIZE_MAX_TAPS");
            return NUS_ERR_UNSUPPORTED;
        }
        return r;
    });
}

int nus_resize_build_axis(int filter, uint32_t in_n, uint32_t out_n, int32_t *left, uint32_t *ntaps, float *weights)
{
    return guarded<int>("nus_resize_build_axis", [&]() -> int {
        if (!left || !ntaps || !weights || in_n == 0 || out_n == 0 || filter < 0 || filter > 2) {
            nus::set_thread_error("nus_resize_build_axis: bad argument");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        const int r = nus::build_resize_axis(static_cast<nus::ResizeFilter>(filter), in_n, out_n, left, ntaps, weights);
        if (r < 0) {
            nus::set_thread_error("nus_resize_build_axis: window exceeds NUS_RESIZE_MAX_TAPS");
            return NUS_ERR_UNSUPPORTED;
        }
        return r;
    });
}

int nus_nearest_build_axis(uint32_t in_n, uint32_t out_n, uint32_t *src)
{
    return guarded<int>("nus_nearest_build_axis", [&]() -> int {
        if (!src || in_n == 0 || out_n == 0) {
            nus::set_thread_error("nus_nearest_build_axis: bad argument");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        nus::build_nearest_axis(in_n, out_n, src);
        return NUS_OK;
    });
}

int nus_bilinear_build_axis(uint32_t in_n, uint32_t out_n, int variant, uint32_t *i0, float *frac)
{
    return guarded<int>("nus_bilinear_build_axis", [&]() -> int {
        if (!i0 || !frac || in_n == 0 || out_n == 0 || (variant != 0 && variant != 1)) {
            nus::set_thread_error("nus_bilinear_build_axis: bad argument");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        nus::build_bilinear_axis(in_n, out_n, variant == 1, i0, frac);
        return NUS_OK;
    });
}

nus_interp *nus_interp_create(int wg_preset)
{
    return guarded<nus_interp *>("nus_interp_create", [&]() -> nus_interp * {
        if (wg_preset < NUS_WG_SQUARE_8X8 || wg_preset > NUS_WG_TALL_8X32) {
            nus::set_thread_error("nus_interp_create: unknown workgroup preset");
            return nullptr;
        }
        return new (std::nothrow) nus_interp(wg_preset);
    });
}

void nus_interp_destroy(nus_interp *h) { delete h; }
int nus_interp_set_device(nus_interp *h, int device)
{
    return guarded<int>("nus_interp_set_device", [&]() -> int { return h ? h->impl.set_device(device) : null_handle(); });
}
int nus_interp_set_input_format(nus_interp *h, int format)
{
    return guarded<int>("nus_interp_set_input_format", [&]() -> int { return h ? h->impl.set_input_format(format) : null_handle(); });
}

int nus_interp_set_mode(nus_interp *h, int mode)
{
    return guarded<int>("nus_interp_set_mode", [&]() -> int { return h ? h->impl.set_mode(mode) : null_handle(); });
}

int nus_interp_mode(const nus_interp *h) { return h ? h->impl.mode() : null_handle(); }

int nus_interp_set_flow_format(nus_interp *h, int format)
{
    return guarded<int>("nus_interp_set_flow_format", [&]() -> int { return h ? h->impl.set_flow_format(format) : null_handle(); });
}

int nus_interp_initialize(nus_interp *h, uint32_t width, uint32_t height)
{
    return guarded<int>("nus_interp_initialize", [&]() -> int { return h ? h->impl.initialize(width, height) : null_handle(); });
}

int nus_interp_interpolate_frames(nus_interp *h, const uint8_t *frame1, size_t len1, const uint8_t *frame2, size_t len2, float t,
                                  uint8_t *out, size_t out_cap)
{
    return guarded<int>("nus_interp_interpolate_frames", [&]() -> int {
        return h ? h->impl.interpolate_frames(frame1, len1, frame2, len2, t, out, out_cap) : null_handle();
    });
}

int nus_interp_set_quality(nus_interp *h, int quality)
{
    return guarded<int>("nus_interp_set_quality", [&]() -> int {
        return h ? h->impl.set_quality(static_cast<nus::InterpolationQuality>(quality)) : null_handle();
    });
}

int nus_interp_quality(const nus_interp *h) { return h ? static_cast<int>(h->impl.quality()) : null_handle(); }
const char *nus_interp_name(const nus_interp *h) { return h ? h->impl.name() : ""; }

int nus_interp_interpolate(nus_interp *h, const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len,
                           const float *flow, uint32_t w, uint32_t hgt, float t, uint8_t *out, size_t out_cap)
{
    return guarded<int>("nus_interp_interpolate", [&]() -> int { return h ? h->impl.interpolate(a, a_len, b, b_len, flow, w, hgt, t, out, out_cap) : null_handle(); });
}

int nus_interp_interpolate_device(nus_interp *h, const void *d_a, size_t a_stride, const void *d_b, size_t b_stride,
                                  const void *d_flow, uint32_t w, uint32_t hgt, float t, void *d_out, uint32_t n_pairs,
                                  void *stream)
{
    return guarded<int>("nus_interp_interpolate_device", [&]() -> int {
        return h ? h->impl.interpolate_device(d_a, a_stride, d_b, b_stride, d_flow, w, hgt, t, d_out, n_pairs,
                                              static_cast<hipStream_t>(stream))
                 : null_handle();
    });
}

int nus_interp_last_gpu_ms(const nus_interp *h, double *ms_out)
{
    return guarded<int>("nus_interp_last_gpu_ms", [&]() -> int {
        if (!h) return null_handle();
        return h->impl.last_gpu_ms(ms_out) ? NUS_OK : NUS_ERR_NOT_INITIALIZED;
    });
}

const char *nus_interp_last_error(const nus_interp *h) { return h ? h->impl.last_error() : "null handle"; }

nus_frame_queue *nus_frame_queue_create(size_t capacity)
{
    return guarded<nus_frame_queue *>("nus_frame_queue_create", [&]() -> nus_frame_queue * { return new (std::nothrow) nus_frame_queue(capacity); });
}
void nus_frame_queue_destroy(nus_frame_queue *q) { delete q; }

int64_t nus_frame_queue_add(nus_frame_queue *q, const uint8_t *rgba, uint32_t w, uint32_t hgt)
{
    return guarded<int64_t>("nus_frame_queue_add", [&]() -> int64_t {
        if (!q) return null_handle();
        if (!rgba || w == 0 || hgt == 0) {
            nus::set_thread_error("nus_frame_queue_add: bad frame");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        return (int64_t)q->impl.add(rgba, w, hgt);
    });
}

static int frame_out(const std::shared_ptr<nus::QueuedFrame> &f, uint8_t *out, size_t out_cap, uint32_t *w,
                     uint32_t *hgt, uint64_t *sequence)
{
    if (!f) return 0;
    if (!out || out_cap < f->data.size()) {
        nus::set_thread_error("frame queue: output buffer too small");
        return NUS_ERR_INVALID_ARGUMENT;
    }
    memcpy(out, f->data.data(), f->data.size());
    if (w) *w = f->width;
    if (hgt) *hgt = f->height;
    if (sequence) *sequence = f->sequence;
    return 1;
}

int nus_frame_queue_latest(nus_frame_queue *q, int64_t timeout_ms, uint8_t *out, size_t out_cap, uint32_t *w,
                           uint32_t *hgt, uint64_t *sequence)
{
    return guarded<int>("nus_frame_queue_latest", [&]() -> int { return q ? frame_out(q->impl.latest(timeout_ms), out, out_cap, w, hgt, sequence) : null_handle(); });
}

int nus_frame_queue_pop(nus_frame_queue *q, int64_t timeout_ms, uint8_t *out, size_t out_cap, uint32_t *w,
                        uint32_t *hgt, uint64_t *sequence)
{
    return guarded<int>("nus_frame_queue_pop", [&]() -> int {
        if (!q) return null_handle();
        // the size check and the removal are one step under the queue's lock: a frame this buffer cannot hold stays queued
        bool too_big = false;
        auto f = q->impl.pop(timeout_ms, out ? out_cap : 0, &too_big);
        if (too_big) {
            nus::set_thread_error("frame queue: output buffer too small");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        return frame_out(f, out, out_cap, w, hgt, sequence);
    });
}

size_t nus_frame_queue_size(const nus_frame_queue *q) { return q ? q->impl.size() : 0; }
size_t nus_frame_queue_capacity(const nus_frame_queue *q) { return q ? q->impl.capacity() : 0; }
uint64_t nus_frame_queue_dropped(const nus_frame_queue *q) { return q ? q->impl.dropped() : 0; }

size_t nus_host_pending_pieces(void) { return nus::parallel_copy_pending(); }

int nus_probe_device(int kind, const void *d_src, void *d_dst, size_t bytes, uint32_t iters, void *stream)
{
    return guarded<int>("nus_probe_device", [&]() -> int {
        auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) % 16) != 0; };
        const bool needs_src = kind == 0 || kind == 1 || kind == 3 || kind == 4;
        if (kind < 0 || kind > 5 || !d_dst || (needs_src && !d_src) || misaligned(d_src) || misaligned(d_dst) || (bytes % 16) ||
            (kind == 5 && iters == 0) || (kind == 4 && (bytes % 1024))) { // kind 4: whole waves only (each writes 4 x 1 KiB)
            nus::set_thread_error("nus_probe_device: bad kind, pointer, size or iteration count");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        if (kind != 5 && bytes == 0) return NUS_OK;
        const hipError_t e = nus::launch_probe(kind, d_src, d_dst, bytes, iters, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            nus::set_thread_error(std::string("HIP error in probe launch: ") + hipGetErrorString(e));
            return NUS_ERR_HIP;
        }
        return NUS_OK;
    });
}

int nus_swizzle_bgra_to_rgba_device(const void *d_in, void *d_out, size_t n_pixels, void *stream)
{
    return guarded<int>("nus_swizzle_bgra_to_rgba_device", [&]() -> int {
        if (!d_in || !d_out || (reinterpret_cast<uintptr_t>(d_in) % 4) || (reinterpret_cast<uintptr_t>(d_out) % 4)) {
            nus::set_thread_error("nus_swizzle_bgra_to_rgba_device: bad pointer");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        if (n_pixels == 0) return NUS_OK;
        const hipError_t e = nus::launch_swizzle_bgra(static_cast<const uint8_t *>(d_in), static_cast<uint8_t *>(d_out),
                                                      n_pixels, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            nus::set_thread_error(std::string("HIP error in swizzle launch: ") + hipGetErrorString(e));
            return NUS_ERR_HIP;
        }
        return NUS_OK;
    });
}

nus_flow *nus_flow_create(void)
{
    return guarded<nus_flow *>("nus_flow_create", [&]() -> nus_flow * { return new (std::nothrow) nus_flow(); });
}
void nus_flow_destroy(nus_flow *h) { delete h; }
int nus_flow_set_device(nus_flow *h, int device)
{
    return guarded<int>("nus_flow_set_device", [&]() -> int { return h ? h->impl.set_device(device) : null_handle(); });
}
int nus_flow_set_tiled(nus_flow *h, int enabled)
{
    return guarded<int>("nus_flow_set_tiled", [&]() -> int { return h ? h->impl.set_tiled(enabled) : null_handle(); });
}
int nus_flow_set_mode(nus_flow *h, int mode)
{
    return guarded<int>("nus_flow_set_mode", [&]() -> int { return h ? h->impl.set_mode(mode) : null_handle(); });
}
int nus_flow_mode(const nus_flow *h) { return h ? h->impl.mode() : NUS_ERR_INVALID_ARGUMENT; }
const char *nus_flow_last_error(const nus_flow *h) { return h ? h->impl.last_error() : "null handle"; }

int nus_flow_rgba8_to_f32(nus_flow *h, const uint8_t *in, uint32_t w, uint32_t hgt, float *out)
{
    return guarded<int>("nus_flow_rgba8_to_f32", [&]() -> int { return h ? h->impl.rgba8_to_f32(in, w, hgt, out) : null_handle(); });
}

int nus_flow_blur(nus_flow *h, const float *in, uint32_t w, uint32_t hgt, float *out)
{
    return guarded<int>("nus_flow_blur", [&]() -> int { return h ? h->impl.blur(in, w, hgt, out) : null_handle(); });
}

int nus_flow_downsample(nus_flow *h, const float *in, uint32_t w, uint32_t hgt, float *out)
{
    return guarded<int>("nus_flow_downsample", [&]() -> int { return h ? h->impl.downsample(in, w, hgt, out) : null_handle(); });
}

int nus_flow_horn_schunck(nus_flow *h, const float *i1, const float *i2, const float *flow_in, uint32_t w, uint32_t hgt,
                          float lambda, uint32_t iterations, float *flow_out)
{
    return guarded<int>("nus_flow_horn_schunck", [&]() -> int { return h ? h->impl.horn_schunck(i1, i2, flow_in, w, hgt, lambda, iterations, flow_out) : null_handle(); });
}

int nus_flow_upsample(nus_flow *h, const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh,
                      float scale)
{
    return guarded<int>("nus_flow_upsample", [&]() -> int { return h ? h->impl.upsample(src, sw, sh, dst, dw, dh, scale) : null_handle(); });
}

int nus_flow_estimate(nus_flow *h, const uint8_t *a, const uint8_t *b, uint32_t w, uint32_t hgt, uint32_t levels,
                      uint32_t coarse_iters, uint32_t refine_iters, float lambda, float *flow_out)
{
    return guarded<int>("nus_flow_estimate", [&]() -> int { return h ? h->impl.estimate(a, b, w, hgt, levels, coarse_iters, refine_iters, lambda, flow_out) : null_handle(); });
}

int nus_flow_estimate_device(nus_flow *h, const void *d_a, const void *d_b, uint32_t w, uint32_t hgt, uint32_t levels,
                             uint32_t coarse_iters, uint32_t refine_iters, float lambda, void *d_flow_out, void *stream)
{
    return guarded<int>("nus_flow_estimate_device", [&]() -> int {
        return h ? h->impl.estimate_device(d_a, d_b, w, hgt, levels, coarse_iters, refine_iters, lambda, d_flow_out,
                                           static_cast<hipStream_t>(stream))
                 : null_handle();
    });
}

int nus_flow_estimate_device_stream(nus_flow *h, const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t hgt,
                                    uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda,
                                    void *d_flows, void *stream)
{
    return guarded<int>("nus_flow_estimate_device_stream", [&]() -> int {
        return h ? h->impl.estimate_device_stream(d_frames, n_frames, w, hgt, levels, coarse_iters, refine_iters, lambda,
                                                  d_flows, static_cast<hipStream_t>(stream))
                 : null_handle();
    });
}

int nus_flow_interpolate_device_stream(nus_flow *h, const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t hgt,
                                       uint32_t levels, uint32_t coarse_iters, uint32_t refine_iters, float lambda, float time_t,
                                       int flow_format, void *d_flows, void *d_mid, void *stream)
{
    return guarded<int>("nus_flow_interpolate_device_stream", [&]() -> int {
        if (flow_format != NUS_FLOW_F32 && flow_format != NUS_FLOW_F16) {
            nus::set_thread_error("nus_flow_interpolate_device_stream: flow_format must be NUS_FLOW_F32 or NUS_FLOW_F16");
            return NUS_ERR_INVALID_ARGUMENT;
        }
        return h ? h->impl.interpolate_device_stream(d_frames, n_frames, w, hgt, levels, coarse_iters, refine_iters, lambda, time_t,
                                                     d_flows, d_mid, static_cast<hipStream_t>(stream), flow_format == NUS_FLOW_F16)
                 : null_handle();
    });
}

} // extern "C"

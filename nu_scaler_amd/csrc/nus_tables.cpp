// nus_tables.cpp -- host-side per-axis tables.  See nus_tables.hpp.
// Compiled with -ffp-contract=off: every f32 operation below rounds on its own.
#include "nus_tables.hpp"

#include <cmath>
#include <cstring>

namespace nus {

void build_nearest_axis(uint32_t in_n, uint32_t out_n, uint32_t *src)
{
    for (uint32_t o = 0; o < out_n; ++o) {
        uint64_t s = ((uint64_t)o * in_n) / out_n;
        src[o] = (uint32_t)(s < in_n - 1 ? s : in_n - 1);
    }
}

void build_bilinear_axis(uint32_t in_n, uint32_t out_n, bool wgsl_form, uint32_t *i0, float *frac)
{
    const float fin = (float)in_n, fout = (float)out_n;
    for (uint32_t o = 0; o < out_n; ++o) {
        float s = (float)o * fin / fout; // multiply, then divide
        if (!wgsl_form) s = std::fmin(s, fin - 1.0f);
        const float fl = wgsl_form ? std::trunc(s) : std::floor(s);
        uint32_t i = (uint32_t)fl;
        if (i > in_n - 1) i = in_n - 1; // only reachable in the wgsl form when out_n < in_n rounds up
        i0[o] = i;
        frac[o] = s - (float)i;
    }
}

namespace {

const float kPi = 3.14159265358979323846f;

float sinc(float t)
{
    const float a = t * kPi;
    if (t == 0.0f) return 1.0f;
    return std::sin(a) / a;
}

float lanczos3(float x)
{
    if (std::fabs(x) < 3.0f) return sinc(x) * sinc(x / 3.0f);
    return 0.0f;
}

// Catmull-Rom: the crate's bicubic_kernel(x, b = 0, c = 0.5)
float catmull_rom(float x)
{
    const float b = 0.0f, c = 0.5f;
    const float a = std::fabs(x);
    float k;
    if (a < 1.0f)
        k = (12.0f - 9.0f * b - 6.0f * c) * (a * a * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
    else if (a < 2.0f)
        k = (-b - 6.0f * c) * (a * a * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
    else
        k = 0.0f;
    return k / 6.0f;
}

float triangle(float x)
{
    const float a = std::fabs(x);
    return a < 1.0f ? 1.0f - a : 0.0f;
}

} // namespace

int build_resize_axis(ResizeFilter filter, uint32_t in_n, uint32_t out_n, int32_t *left, uint32_t *ntaps, float *weights)
{
    if (in_n == 0 || out_n == 0) return -1;
    float (*kernel)(float) = lanczos3;
    float support = 3.0f;
    if (filter == ResizeFilter::CatmullRom) {
        kernel = catmull_rom;
        support = 2.0f;
    } else if (filter == ResizeFilter::Triangle) {
        kernel = triangle;
        support = 1.0f;
    }
    const float ratio = (float)in_n / (float)out_n;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = support * sratio;
    int worst = 0;
    for (uint32_t o = 0; o < out_n; ++o) {
        float *ws = weights + (size_t)o * kResizeMaxTaps;
        std::memset(ws, 0, sizeof(float) * kResizeMaxTaps);
        // centre of output pixel o in input coordinates (half-pixel convention)
        float centre = ((float)o + 0.5f) * ratio;
        int64_t lo = (int64_t)std::floor(centre - src_support);
        if (lo < 0) lo = 0;
        if (lo > (int64_t)in_n - 1) lo = (int64_t)in_n - 1;
        int64_t hi = (int64_t)std::ceil(centre + src_support);
        if (hi < lo + 1) hi = lo + 1;
        if (hi > (int64_t)in_n) hi = (int64_t)in_n;
        centre = centre - 0.5f;
        const uint32_t n = (uint32_t)(hi - lo);
        left[o] = (int32_t)lo;
        ntaps[o] = n;
        if (n > kResizeMaxTaps) return -1;
        float sum = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float w = kernel(((float)(lo + (int64_t)i) - centre) / sratio);
            ws[i] = w;
            sum += w;
        }
        for (uint32_t i = 0; i < n; ++i) ws[i] /= sum;
        if ((int)n > worst) worst = (int)n;
    }
    return worst;
}

void build_axis_tables(uint32_t in_n, uint32_t out_n, bool wgsl_form, AxisTables &t, ResizeFilter filter)
{
    t.filter = filter;
    t.in_n = in_n;
    t.out_n = out_n;
    t.nn_src.resize(out_n);
    t.bl_i0.resize(out_n);
    t.bl_frac.resize(out_n);
    t.lz_left.resize(out_n);
    t.lz_ntaps.resize(out_n);
    t.lz_w.assign((size_t)out_n * kResizeMaxTaps, 0.0f);
    build_nearest_axis(in_n, out_n, t.nn_src.data());
    build_bilinear_axis(in_n, out_n, wgsl_form, t.bl_i0.data(), t.bl_frac.data());
    t.lz_max_taps = build_resize_axis(filter, in_n, out_n, t.lz_left.data(), t.lz_ntaps.data(), t.lz_w.data());
}

bool lanczos_x2_phase_frame(const AxisTables &t, std::vector<float> &w6)
{
    if (t.out_n != 2 * t.in_n || t.lz_max_taps < 0) return false;
    w6.assign((size_t)t.out_n * 6, 0.0f);
    for (uint32_t o = 0; o < t.out_n; ++o) {
        const int32_t base = (int32_t)(o >> 1) - 3 + (int32_t)(o & 1);
        const float *ws = t.lz_w.data() + (size_t)o * kResizeMaxTaps;
        for (uint32_t i = 0; i < t.lz_ntaps[o]; ++i) {
            const int32_t j = t.lz_left[o] + (int32_t)i - base;
            if (j < 0 || j >= 6) {
                if (ws[i] != 0.0f) return false;
                continue;
            }
            w6[(size_t)o * 6 + j] = ws[i];
        }
    }
    return true;
}

bool lanczos_x2_interior_uniform(const AxisTables &t, const std::vector<float> &w6)
{
    if (t.in_n < 16) return false;
    // outputs 2k, 2k+1 with 4 <= k <= in_n - 5 have all taps inside the image
    for (uint32_t o = 8; o + 8 < t.out_n; ++o) {
        const float *ref = w6.data() + (size_t)(8 + (o & 1)) * 6;
        if (std::memcmp(ref, w6.data() + (size_t)o * 6, 6 * sizeof(float)) != 0) return false;
    }
    return true;
}

bool lanczos_xs_phase_frame(const AxisTables &t, uint32_t S, std::vector<float> &w6)
{
    if (S < 2 || t.out_n != S * t.in_n || t.lz_max_taps < 0) return false;
    w6.assign((size_t)t.out_n * 6, 0.0f);
    for (uint32_t o = 0; o < t.out_n; ++o) {
        const uint32_t p = o % S;
        const int32_t base = (int32_t)(o / S) - 3 + (2 * p + 1 > S ? 1 : 0);
        const float *ws = t.lz_w.data() + (size_t)o * kResizeMaxTaps;
        for (uint32_t i = 0; i < t.lz_ntaps[o]; ++i) {
            const int32_t j = t.lz_left[o] + (int32_t)i - base;
            if (j < 0 || j >= 6) {
                if (ws[i] != 0.0f) return false;
                continue;
            }
            w6[(size_t)o * 6 + j] = ws[i];
        }
    }
    return true;
}

bool lanczos_xs_interior_uniform(const AxisTables &t, uint32_t S, const std::vector<float> &w6)
{
    if (t.in_n < 16) return false;
    for (uint32_t o = 4 * S; o + 4 * S < t.out_n; ++o) {
        const float *ref = w6.data() + (size_t)(8 * S + o % S) * 6;
        if (std::memcmp(ref, w6.data() + (size_t)o * 6, 6 * sizeof(float)) != 0) return false;
    }
    return true;
}

bool lanczos_xs_weight_classes(const AxisTables &t, uint32_t S, const std::vector<float> &w6, bool lanes,
                               std::vector<uint32_t> &cls, std::vector<float> &classes)
{
    if (t.in_n < 16 || (lanes && (t.in_n % 4) != 0)) return false;
    const size_t frame = (size_t)S * 6;
    cls.assign(t.in_n, 0);
    classes.clear();
    for (uint32_t k = 4; k + 4 < t.in_n; ++k) {
        const float *w = w6.data() + (size_t)k * frame; // outputs S k .. S k + S - 1
        uint32_t c = 0;
        const uint32_t n = (uint32_t)(classes.size() / frame);
        for (; c < n; ++c)
            if (std::memcmp(classes.data() + (size_t)c * frame, w, frame * sizeof(float)) == 0) break;
        if (c == n) {
            if (n == kXsMaxClasses) return false;
            classes.insert(classes.end(), w, w + frame);
        }
        cls[k] = c;
    }
    for (uint32_t k = 4; lanes && k + 8 <= t.in_n; k += 4) // a lane's 4 columns share a class
        if (cls[k] != cls[k + 1] || cls[k] != cls[k + 2] || cls[k] != cls[k + 3]) return false;
    return !classes.empty();
}

bool lanczos_r32_phase_frame(const AxisTables &t, std::vector<float> &w6)
{
    if ((t.in_n & 1) != 0 || 2 * (uint64_t)t.out_n != 3 * (uint64_t)t.in_n || t.lz_max_taps < 0) return false;
    w6.assign((size_t)t.out_n * 6, 0.0f);
    for (uint32_t o = 0; o < t.out_n; ++o) {
        const int32_t base = 2 * (int32_t)(o / 3) - 3 + (int32_t)(o % 3);
        const float *ws = t.lz_w.data() + (size_t)o * kResizeMaxTaps;
        for (uint32_t i = 0; i < t.lz_ntaps[o]; ++i) {
            const int32_t j = t.lz_left[o] + (int32_t)i - base;
            if (j < 0 || j >= 6) {
                if (ws[i] != 0.0f) return false;
                continue;
            }
            w6[(size_t)o * 6 + j] = ws[i];
        }
    }
    return true;
}

bool lanczos_r43_phase_frame(const AxisTables &t, std::vector<float> &w6)
{
    if ((t.in_n % 3) != 0 || 3 * (uint64_t)t.out_n != 4 * (uint64_t)t.in_n || t.lz_max_taps < 0) return false;
    w6.assign((size_t)t.out_n * 6, 0.0f);
    for (uint32_t o = 0; o < t.out_n; ++o) {
        const int32_t base = 3 * (int32_t)(o / 4) - 3 + (int32_t)(o % 4);
        const float *ws = t.lz_w.data() + (size_t)o * kResizeMaxTaps;
        for (uint32_t i = 0; i < t.lz_ntaps[o]; ++i) {
            const int32_t j = t.lz_left[o] + (int32_t)i - base;
            if (j < 0 || j >= 6) {
                if (ws[i] != 0.0f) return false;
                continue;
            }
            w6[(size_t)o * 6 + j] = ws[i];
        }
    }
    return true;
}

bool lanczos_r43_interior_uniform(const AxisTables &t, const std::vector<float> &w6)
{
    if (t.in_n < 18) return false;
    for (uint32_t o = 8; o + 8 < t.out_n; ++o) {
        const float *ref = w6.data() + (size_t)(8 + o % 4) * 6;
        if (std::memcmp(ref, w6.data() + (size_t)o * 6, 6 * sizeof(float)) != 0) return false;
    }
    return true;
}

bool lanczos_pq_phase_frame(const AxisTables &t, uint32_t P, uint32_t Q, std::vector<float> &w6)
{
    if (P <= Q || Q == 0 || (t.in_n % Q) != 0 || (uint64_t)Q * t.out_n != (uint64_t)P * t.in_n || t.lz_max_taps < 0) return false;
    w6.assign((size_t)t.out_n * 6, 0.0f);
    for (uint32_t o = 0; o < t.out_n; ++o) {
        const int64_t p = o % P, num = (2 * p + 1) * (int64_t)Q - 7 * (int64_t)P, den = 2 * (int64_t)P;
        const int64_t fl = num >= 0 ? num / den : -((-num + den - 1) / den);
        const int32_t base = (int32_t)((int64_t)Q * (o / P) + fl + 1);
        const float *ws = t.lz_w.data() + (size_t)o * kResizeMaxTaps;
        for (uint32_t i = 0; i < t.lz_ntaps[o]; ++i) {
            const int32_t j = t.lz_left[o] + (int32_t)i - base;
            if (j < 0 || j >= 6) {
                if (ws[i] != 0.0f) return false;
                continue;
            }
            w6[(size_t)o * 6 + j] = ws[i];
        }
    }
    return true;
}

bool lanczos_r32_weight_classes(const AxisTables &t, const std::vector<float> &w6, bool lanes, std::vector<uint32_t> &cls,
                                std::vector<float> &classes)
{
    if (t.in_n < 16 || (lanes && (t.in_n % 4) != 0)) return false;
    const uint32_t pairs = t.in_n / 2;
    const size_t frame = 3 * 6;
    cls.assign(pairs, 0);
    classes.clear();
    for (uint32_t g = 2; g + 2 < pairs; ++g) {
        const float *w = w6.data() + (size_t)g * frame; // outputs 3 g .. 3 g + 2
        uint32_t c = 0;
        const uint32_t n = (uint32_t)(classes.size() / frame);
        for (; c < n; ++c)
            if (std::memcmp(classes.data() + (size_t)c * frame, w, frame * sizeof(float)) == 0) break;
        if (c == n) {
            if (n == kXsMaxClasses) return false;
            classes.insert(classes.end(), w, w + frame);
        }
        cls[g] = c;
    }
    for (uint32_t g = 2; lanes && g + 4 <= pairs; g += 2) // a lane's columns 2g .. 2g+3 share a class
        if (cls[g] != cls[g + 1]) return false;
    return !classes.empty();
}

namespace {

const uint32_t kMagic = 0x4C53554Eu; // "NUSL"

template <typename T>
void put(std::vector<uint8_t> &b, const T *p, size_t n)
{
    const uint8_t *s = reinterpret_cast<const uint8_t *>(p);
    b.insert(b.end(), s, s + n * sizeof(T));
}

void put_axis(std::vector<uint8_t> &b, const AxisTables &t)
{
    const uint32_t hdr[4] = {t.in_n, t.out_n, (uint32_t)t.lz_max_taps, (uint32_t)t.filter};
    put(b, hdr, 4);
    put(b, t.nn_src.data(), t.out_n);
    put(b, t.bl_i0.data(), t.out_n);
    put(b, t.bl_frac.data(), t.out_n);
    put(b, t.lz_left.data(), t.out_n);
    put(b, t.lz_ntaps.data(), t.out_n);
    put(b, t.lz_w.data(), (size_t)t.out_n * kResizeMaxTaps);
}

template <typename T>
bool get(const uint8_t *&p, const uint8_t *end, T *dst, size_t n)
{
    if ((size_t)(end - p) < n * sizeof(T)) return false;
    std::memcpy(dst, p, n * sizeof(T));
    p += n * sizeof(T);
    return true;
}

bool get_axis(const uint8_t *&p, const uint8_t *end, AxisTables &t)
{
    uint32_t hdr[4];
    if (!get(p, end, hdr, 4)) return false;
    if (hdr[1] == 0 || hdr[1] > (1u << 24) || hdr[3] > 2u) return false;
    t.in_n = hdr[0];
    t.out_n = hdr[1];
    t.lz_max_taps = (int)hdr[2];
    t.filter = static_cast<ResizeFilter>(hdr[3]);
    t.nn_src.resize(t.out_n);
    t.bl_i0.resize(t.out_n);
    t.bl_frac.resize(t.out_n);
    t.lz_left.resize(t.out_n);
    t.lz_ntaps.resize(t.out_n);
    t.lz_w.resize((size_t)t.out_n * kResizeMaxTaps);
    return get(p, end, t.nn_src.data(), t.out_n) && get(p, end, t.bl_i0.data(), t.out_n) &&
           get(p, end, t.bl_frac.data(), t.out_n) && get(p, end, t.lz_left.data(), t.out_n) &&
           get(p, end, t.lz_ntaps.data(), t.out_n) && get(p, end, t.lz_w.data(), (size_t)t.out_n * kResizeMaxTaps);
}

} // namespace

bool build_down_stream_tables(const AxisTables &t, std::vector<uint32_t> &rows, std::vector<int32_t> &done_row)
{
    if (t.lz_max_taps <= 0 || t.out_n == 0 || t.out_n >= (1u << 28)) return false;
    const uint32_t n = t.in_n, extra_cap = 2 * kDownSlots;
    rows.assign((size_t)(n + extra_cap) * 8, 0u);
    for (size_t r = 0; r < (size_t)n + extra_cap; ++r) rows[r * 8 + 7] = kDownNone;
    done_row.assign(t.out_n, -1);
    std::vector<int64_t> slot_free_from(kDownSlots, 0); // first input row a new window may occupy the slot from
    uint32_t extra = 0;
    for (uint32_t y = 0; y < t.out_n; ++y) {
        const uint32_t slot = y % kDownSlots;
        const int64_t left = t.lz_left[y], taps = t.lz_ntaps[y], end = left + taps - 1;
        if (taps <= 0 || left < slot_free_from[slot] || end >= (int64_t)n) return false;
        slot_free_from[slot] = end + 1;
        for (int64_t k = 0; k < taps; ++k) {
            const float w = t.lz_w[(size_t)y * kResizeMaxTaps + (size_t)k];
            memcpy(&rows[(size_t)(left + k) * 8 + slot], &w, sizeof(w));
        }
        size_t at = (size_t)end;
        if (rows[at * 8 + 7] != kDownNone) { // a second window ending on this row: only the far border does that
            if (end != (int64_t)n - 1 || extra >= extra_cap) return false;
            at = (size_t)n + extra++;
        }
        rows[at * 8 + 7] = (slot << 28) | y;
        done_row[y] = (int32_t)at;
    }
    rows.resize((size_t)(n + extra) * 8);
    return true;
}

std::vector<uint8_t> serialize_tables(const AxisTables &x, const AxisTables &y)
{
    std::vector<uint8_t> b;
    const uint32_t hdr[2] = {kMagic, 2u};
    put(b, hdr, 2);
    put_axis(b, x);
    put_axis(b, y);
    return b;
}

bool deserialize_tables(const uint8_t *buf, size_t len, AxisTables &x, AxisTables &y, std::string &err)
{
    const uint8_t *p = buf, *end = buf + len;
    uint32_t hdr[2];
    if (!get(p, end, hdr, 2) || hdr[0] != kMagic || hdr[1] != 2u) {
        err = "table blob: bad magic or version";
        return false;
    }
    if (!get_axis(p, end, x) || !get_axis(p, end, y) || p != end) {
        err = "table blob: truncated or oversized";
        return false;
    }
    // indices must stay inside the source axis: the kernels trust them
    for (AxisTables *t : {&x, &y}) {
        uint32_t max_taps = 0;
        for (uint32_t o = 0; o < t->out_n; ++o) {
            if (t->nn_src[o] >= t->in_n || t->bl_i0[o] >= t->in_n) {
                err = "table blob: index out of range";
                return false;
            }
            if (t->lz_max_taps >= 0) {
                if (o == 0) max_taps = 0;
                max_taps = t->lz_ntaps[o] > max_taps ? t->lz_ntaps[o] : max_taps;
                if (t->lz_left[o] < 0 || t->lz_ntaps[o] == 0 || t->lz_ntaps[o] > kResizeMaxTaps ||
                    (uint64_t)t->lz_left[o] + t->lz_ntaps[o] > t->in_n) {
                    err = "table blob: tap window out of range";
                    return false;
                }
                // the segment kernels take a segment's footprint from its first and last output
                if (o > 0 && (t->lz_left[o] < t->lz_left[o - 1] ||
                              t->lz_left[o] + (int64_t)t->lz_ntaps[o] < t->lz_left[o - 1] + (int64_t)t->lz_ntaps[o - 1])) {
                    err = "table blob: tap windows must not move backwards";
                    return false;
                }
            }
        }
        // the widest window drives the kernel choice: take it from the windows themselves, not from the blob's header
        if (t->lz_max_taps >= 0) t->lz_max_taps = (int32_t)max_taps;
    }
    return true;
}

} // namespace nus

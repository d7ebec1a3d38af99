// nus_tables.hpp -- host-side per-axis tables (computed once at initialize).
// The kernels never divide or call sinf: every index, fraction and filter weight is
// produced here with plain IEEE f32 host arithmetic, so it is the same value the CPU
// algorithm uses, and identical on every GPU of a node (tables can be exported and
// broadcast, see nus_upscaler_export_tables).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace nus {

constexpr uint32_t kResizeMaxTaps = 32; // == NUS_RESIZE_MAX_TAPS

// nearest: src[o] = min(o * in_n / out_n, in_n - 1)   (Nu_scale/src/upscale/common.rs:191-192)
void build_nearest_axis(uint32_t in_n, uint32_t out_n, uint32_t *src);

// bilinear: i0[o], frac[o].  cpu form clamps the coordinate to in_n - 1
// (Nu_scale/src/upscale/common.rs:204-215); wgsl form does not (upscale/mod.rs:241-248).
void build_bilinear_axis(uint32_t in_n, uint32_t out_n, bool wgsl_form, uint32_t *i0, float *frac);

// Separable resampling filters of image-0.24.9 `imageops::resize` (FilterType::*).
enum class ResizeFilter : int { Lanczos3 = 0, CatmullRom = 1, Triangle = 2 };

// Tap windows, image-0.24.9 vertical_sample / horizontal_sample convention.
// weights is out_n * kResizeMaxTaps, zero padded.  Returns max ntaps, or -1 when a
// window needs more than kResizeMaxTaps taps (Lanczos-3: down-scaling by more than ~4.8x).
int build_resize_axis(ResizeFilter filter, uint32_t in_n, uint32_t out_n, int32_t *left, uint32_t *ntaps, float *weights);
inline int build_lanczos3_axis(uint32_t in_n, uint32_t out_n, int32_t *left, uint32_t *ntaps, float *weights)
{
    return build_resize_axis(ResizeFilter::Lanczos3, in_n, out_n, left, ntaps, weights);
}

struct AxisTables {
    uint32_t in_n = 0, out_n = 0;
    std::vector<uint32_t> nn_src;
    std::vector<uint32_t> bl_i0;
    std::vector<float> bl_frac;
    ResizeFilter filter = ResizeFilter::Lanczos3; // which filter the lz_* tables were built for
    std::vector<int32_t> lz_left;
    std::vector<uint32_t> lz_ntaps;
    std::vector<float> lz_w; // out_n * kResizeMaxTaps
    int lz_max_taps = 0;     // -1: unsupported ratio
};

void build_axis_tables(uint32_t in_n, uint32_t out_n, bool wgsl_form, AxisTables &t,
                       ResizeFilter filter = ResizeFilter::Lanczos3);

// Exact-x2 view of a Lanczos axis: weights of output o in its 6-tap phase frame
// (base = (o>>1) - 3 + (o&1)).  Returns false if any non-zero tap falls outside
// the frame (then the x2 kernel must not be used).
bool lanczos_x2_phase_frame(const AxisTables &t, std::vector<float> &w6);

// True when every interior output (taps untouched by the image border) has the same
// phase-frame weights as outputs 8 (even) and 9 (odd).
bool lanczos_x2_interior_uniform(const AxisTables &t, const std::vector<float> &w6);

// Integer factor S (out_n == S * in_n): the same frames for output o = S k + p, starting at
// k - 3 + delta_p with delta_p = (2 p + 1 > S) (the output centre lies right of input pixel k).
bool lanczos_xs_phase_frame(const AxisTables &t, uint32_t S, std::vector<float> &w6);
// True when every interior output (k in [4, in_n - 5]) has the weights of output S * 8 + p of its phase.
bool lanczos_xs_interior_uniform(const AxisTables &t, uint32_t S, const std::vector<float> &w6);
// When they do not (x3: (o + 0.5) * fl(1/3) is rounded in f32, so the fractional position of a sample -- and with it
// every weight -- moves with the binade of the coordinate): group the interior input indices by the exact bits of
// their S phase frames.  cls[k] = class of input index k (0 for the border indices, whose weights the kernels take
// elsewhere), classes[c][p][j] = the weights.  False if there are more than kXsMaxClasses classes or, with `lanes`
// (the horizontal axis), if one of the aligned groups of 4 input indices (one lane's columns) spans two classes.
constexpr uint32_t kXsMaxClasses = 16;
bool lanczos_xs_weight_classes(const AxisTables &t, uint32_t S, const std::vector<float> &w6, bool lanes,
                               std::vector<uint32_t> &cls, std::vector<float> &classes);

// Factor 3/2 (2 out_n == 3 in_n, in_n even): output o = 3 g + p belongs to the input pair g = (2g, 2g+1) and its taps
// lie in the 6-slot frame that starts at 2 g - 3 + p (the output centres sit at 2g - 1/6, 2g + 1/2, 2g + 7/6).
bool lanczos_r32_phase_frame(const AxisTables &t, std::vector<float> &w6);
// Weight classes as lanczos_xs_weight_classes, per input PAIR: cls[g], classes[c][p][j] (3 phases).  `lanes`: the two
// pairs of an aligned group of 4 input indices (one lane's columns) must share a class.  Pairs 0, 1 and the last two
// (frames cut by the border) are class 0 and not compared.
bool lanczos_r32_weight_classes(const AxisTables &t, const std::vector<float> &w6, bool lanes, std::vector<uint32_t> &cls,
                                std::vector<float> &classes);

// Factor 4/3 (3 out_n == 4 in_n): output o = 4 g + p belongs to the input group g = (3g .. 3g+2), frame start 3 g - 3 + p.
// The ratio 3/4 and every sample centre are exact in f32: the interior outputs of a phase share one set of weights.
bool lanczos_r43_phase_frame(const AxisTables &t, std::vector<float> &w6);
// True when every output of the groups 2 .. in_n / 3 - 3 has the weights of output 8 + p of its phase.
bool lanczos_r43_interior_uniform(const AxisTables &t, const std::vector<float> &w6);
// Small rational factors P/Q (Q out_n == P in_n, P > Q; nus_k_lanczos_pq.hip): output o = P g + p belongs to the input group
// g = (Q g .. Q g + Q - 1), frame start Q g + floor(((2 p + 1) Q - 7 P) / 2 P) + 1.  No uniformity is asked for: the kernel takes
// every output's weights from w6.  False when a non-zero weight falls outside the frame.
bool lanczos_pq_phase_frame(const AxisTables &t, uint32_t P, uint32_t Q, std::vector<float> &w6);

// Down-scaling stream tables for k_resize_down (7 accumulator slots, slot of output y = y % 7).
// rows: (in_n + extra) x 8 words -- per input row the f32 weight it carries in each slot (0 where the row is
// outside the window of the output that owns the slot), then one completion word: 0xFFFFFFFF, or (slot << 28 | y)
// of the output whose window ends on this row.  Windows cut by the far border end together on the last input
// row; all but the first get a pseudo-row of zero weights behind it, so every row completes at most one output.
// done_row[y]: the (pseudo-)row on which y completes.  False if some window would need an occupied slot.
constexpr uint32_t kDownSlots = 7, kDownNone = 0xFFFFFFFFu;
bool build_down_stream_tables(const AxisTables &t, std::vector<uint32_t> &rows, std::vector<int32_t> &done_row);

// Serialisation for the multi-GPU LUT broadcast.
std::vector<uint8_t> serialize_tables(const AxisTables &x, const AxisTables &y);
bool deserialize_tables(const uint8_t *buf, size_t len, AxisTables &x, AxisTables &y, std::string &err);

} // namespace nus

// nus_flow.hpp -- optical-flow front end of the frame interpolator ("next" row, SURVEY.md
// section 8f rank 1): mirrors WgpuFrameInterpolator::build_pyramid / compute_coarse_flow
// (nu_scaler_core/src/wgpu_interpolator.rs:969-1203) on HIP, plus the coarse-to-fine
// warm start the reference sketches but never wires (its refine path is dead code).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

namespace nus {

class HipFlowEstimator {
public:
    HipFlowEstimator() = default;
    ~HipFlowEstimator();
    HipFlowEstimator(const HipFlowEstimator &) = delete;
    HipFlowEstimator &operator=(const HipFlowEstimator &) = delete;

    int set_device(int device);
    // true (default): derivatives once per level + K Jacobi steps per launch in LDS; false: one
    // plain kernel per step (the shader's structure).  Bit-identical results.
    int set_tiled(int mode); // 0 plain per-step kernel, 1 multi-step kernel chosen by size, 2 LDS tiles, 3 streamed
    const char *last_error() const { return error_.c_str(); }

    // Primitives on host buffers (parity tests, integration).  f32 RGBA images, float2 flows.
    int rgba8_to_f32(const uint8_t *in, uint32_t w, uint32_t h, float *out);
    int blur(const float *in, uint32_t w, uint32_t h, float *out);        // H pass then V pass
    int downsample(const float *in, uint32_t w, uint32_t h, float *out);  // -> ((w+1)/2, (h+1)/2)
    int horn_schunck(const float *i1, const float *i2, const float *flow_in_or_null, uint32_t w, uint32_t h,
                     float lambda, uint32_t iterations, float *flow_out);
    int upsample(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh, float scale);

    // Full estimator: RGBA8 frames -> dense flow (w*h*2 floats, pixel delta A -> B).
    int estimate(const uint8_t *a, const uint8_t *b, uint32_t w, uint32_t h, uint32_t levels, uint32_t coarse_iters,
                 uint32_t refine_iters, float lambda, float *flow_out);
    // 0 EXACT (default): every stage bit-identical to the oracle's restatement of the shaders; 1 FAST: the Jacobi steps of the
    // estimators (estimate, estimate_device, estimate_device_stream) in separable sums / reciprocals / FMAs -- flow within 1e-3 px.
    // The primitives (blur, downsample, horn_schunck, upsample) are always exact.
    int set_mode(int mode);
    int mode() const { return fast_ ? 1 : 0; }
    int estimate_device(const void *d_a, const void *d_b, uint32_t w, uint32_t h, uint32_t levels,
                        uint32_t coarse_iters, uint32_t refine_iters, float lambda, void *d_flow_out,
                        hipStream_t stream);
    // n_frames consecutive RGBA8 frames -> n_frames - 1 flows (k -> k+1), each pyramid built once.
    int estimate_device_stream(const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels,
                               uint32_t coarse_iters, uint32_t refine_iters, float lambda, void *d_flows,
                               hipStream_t stream);
    // The reference's intended interpolate() as ONE pipeline (wgpu_interpolator.rs:881-935: pyramid -> coarse flow -> warp): the
    // flows of estimate_device_stream AND the n_frames - 1 in-between frames at time t warped + blended with them (dense-flow warp
    // in FMA mode) into d_mid: the warp kernel runs behind the estimator on the flow where it is (the caller's buffer, or the
    // workspace when d_flows == nullptr).  (NUS_HS_FUSED_WARP=1: the finest level's last Jacobi launch warps with the flow it has
    // just finished instead -- HsWarp; same bytes, measured slower, off by default; tests/test_flow.py runs both.)
    int interpolate_device_stream(const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels,
                                  uint32_t coarse_iters, uint32_t refine_iters, float lambda, float t, void *d_flows, void *d_mid,
                                  hipStream_t stream, bool flow_half = false);

private:
    struct Pyramid { // level geometry; levels are packed at `offset` (16 bytes per pixel reserved)
        uint32_t levels = 0, w[12] = {0}, h[12] = {0};
        size_t offset[12] = {0}, total = 0;
    };
    int plan(uint32_t w, uint32_t h, uint32_t levels, Pyramid &g); // geometry + workspace
    int build_pyramid(const void *frame, int pyr_slot, const Pyramid &g, hipStream_t stream);
    int solve(int slot_a, int slot_b, const Pyramid &g, uint32_t coarse_iters, uint32_t refine_iters, float lambda,
              void *d_flow_out, hipStream_t stream);
    // multi-step kernels, a chunk of consecutive pairs per launch (pairs on the grid's y / z axis): as many as fit the
    // workspace budget (64 MB per 1080p pair, 88 MB with coefficient planes), at most 150 (round 6; 100 and 6 GiB before: with
    // more pairs per launch the levels are cut into fewer row blocks, i.e. fewer halo rows are recomputed -- the motion step with
    // chunks of 150 units 19.5 - 19.7 against 20.0 - 20.4 ms per 300 units, profiles/r06_flow_chunk_size.txt; 12 GiB of the
    // GPU's 288);  64 -> 100 pairs per chunk had been 76 -> 71 us per pair (profiles/r02_flow_jacobi_streamed_ab.txt)
    static constexpr uint32_t kStreamMaxChunkPairs = 150;
    static constexpr size_t kStreamWorkspaceBytes = (size_t)12 << 30;
    int stream_impl(const void *d_frames, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t levels, uint32_t coarse_iters,
                    uint32_t refine_iters, float lambda, void *d_flows, void *d_mid, float t, hipStream_t stream, bool flow_half = false);
    int solve_batch(const uint8_t *d_frames, uint32_t pairs, const Pyramid &g, uint32_t coarse_iters, uint32_t refine_iters,
                    float lambda, uint8_t *d_flows, hipStream_t stream, uint8_t *d_mid = nullptr, float t = 0.5f,
                    bool flow_half = false);
    int fail(int status, const std::string &msg);
    int fail_hip(hipError_t e, const char *what);
    int ensure_device();
    int reserve(size_t bytes, int slot); // grow-only device scratch slots
    void release();

    std::mutex mu_;
    int device_ = 0;
    bool ready_ = false;
    bool tiled_ = true;
    int jacobi_ = 0; // JacobiKernel
    bool fast_ = false; // set_mode(1): the estimator's Jacobi steps in FAST arithmetic (k_hs_stream_fast), every level streamed
    hipStream_t stream_ = nullptr;
    static constexpr int kSlotCount = 11; // 0-5 pyramids / flows / planes, 6-7 the host entry point's frames, 8 the FAST pair, 9 one pair's flow, 10 a chunk's flows as f16
    void *slot_[kSlotCount] = {nullptr};
    size_t slot_cap_[kSlotCount] = {0};
    std::string error_;
};

} // namespace nus

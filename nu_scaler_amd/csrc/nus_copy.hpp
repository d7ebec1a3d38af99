// nus_copy.hpp -- helper threads for the staging copies of the host path (HipUpscaler::upscale[_batch],
// HipFrameInterpolator::interpolate).  The reference's host path makes three single-threaded copies of every
// 33 MB output (upscale/mod.rs:1040-1058, lib.rs:111); here one copy out of the pinned staging buffer remains,
// and at ~20 GB/s per core it, not PCIe, is what bounds the frames per second of pageable callers -- so it is
// split over a few process-wide worker threads.
#pragma once

#include <cstddef>

namespace nus {

// memcpy(dst, src, bytes), large copies split over the pool's workers and the calling thread.  Safe to call
// from several threads (a second caller copies on its own thread); NUS_COPY_THREADS=0 turns the workers off.
void parallel_copy(void *dst, const void *src, size_t bytes);

// number of worker threads in use (0 before the first large copy or when disabled)
int parallel_copy_workers();

} // namespace nus

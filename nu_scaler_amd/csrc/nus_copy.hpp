// nus_copy.hpp -- helper threads for the staging copies of the host path (HipUpscaler::upscale[_batch],
// HipFrameInterpolator::interpolate).  The reference's host path makes three single-threaded copies of every
// 33 MB output (upscale/mod.rs:1040-1058, lib.rs:111); here one copy out of the pinned staging buffer remains,
// and at ~20 GB/s per core it, not PCIe, is what bounds the frames per second of pageable callers -- so it is
// split over a few process-wide worker threads.
#pragma once

#include <cstddef>

namespace nus {

// memcpy(dst, src, bytes), large copies split over the pool's workers and the calling thread.  Safe to call
// from several threads (their pieces share one queue); NUS_COPY_THREADS=0 turns the workers off.
void parallel_copy(void *dst, const void *src, size_t bytes);

// The same in two halves, for callers that keep several copies in flight (upscale_batch: the staging copy of frame i+1 next
// to the copy-out of frame i): parallel_copy_async queues the copy's pieces for the workers and returns; the bytes have
// been copied when parallel_copy_wait(ticket) returns (the waiting thread works on queued pieces meanwhile).  A ticket may
// collect several copies; both buffers of each must stay valid until the wait returns.  The ticket's count belongs to the pool.
struct CopyTicket {
    size_t left = 0; // pieces not yet copied (guarded by the pool's mutex)
};
void parallel_copy_async(void *dst, const void *src, size_t bytes, CopyTicket &ticket);
void parallel_copy_wait(CopyTicket &ticket);

// Ask the workers to make the pages of a (pageable, writable) buffer present while something else is going on -- the output
// buffer of a frame while the frame is on the GPU: a result buffer fresh from the allocator otherwise takes its first-touch
// faults inside the copy-out.  Contents are never changed; already-present pages cost a page-table walk.  Nothing is queued
// for small buffers or when the pool has no workers.  Wait for `ticket` (parallel_copy_wait) before the buffer may go away.
void parallel_populate_async(void *dst, size_t bytes, CopyTicket &ticket);

// number of worker threads in use (0 when disabled or in a forked child)
int parallel_copy_workers();

} // namespace nus

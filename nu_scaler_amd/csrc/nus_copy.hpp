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
    bool low = false; // holds populate requests (the pool's low-priority queue)
};
void parallel_copy_async(void *dst, const void *src, size_t bytes, CopyTicket &ticket);
void parallel_copy_wait(CopyTicket &ticket);

// The pages of a (pageable, writable) output buffer made present while something else is going on -- while its frame is staged,
// uploaded, computed and on the wire back: a result buffer fresh from the allocator (the Vec / PyBytes the trait's `upscale`
// returns) otherwise takes its first-touch faults inside the copy-out.  Two steps:
//   parallel_populate_prepare  (calling thread, BEFORE any request for other buffers of the same call is queued where possible)
//       false: nothing to do -- the buffer is small, resident already (sampled with mincore), or the pool has no workers;
//       true: the mapping is fresh; it has been given the transparent-huge-page hint and wants populate requests;
//   parallel_populate_async    queues the requests in the pool's LOW-priority queue (idle workers take them; a thread waiting
//       for a copy never does).  Contents are never changed.  Wait for `ticket` (parallel_copy_wait) before the buffer may go away.
bool parallel_populate_prepare(void *dst, size_t bytes);
void parallel_populate_async(void *dst, size_t bytes, CopyTicket &ticket);

// What parallel_populate_prepare asks before it leaves a transparent-huge-page hint on a buffer: is the block its own mapping (at
// least 32 MiB, or carrying glibc's mmapped-chunk header), so that the hint is unmapped with it?  (Exposed for the host tests.)
bool parallel_populate_own_mapping(const void *p, size_t bytes);

// pieces queued or being worked on, over all tickets (0 when no host call is in progress)
size_t parallel_copy_pending();

// number of worker threads in use (0 when disabled or in a forked child)
int parallel_copy_workers();

} // namespace nus

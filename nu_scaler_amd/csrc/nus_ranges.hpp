// nus_ranges.hpp -- what this library has told the HIP runtime or the kernel about HOST memory: ranges it registered for DMA
// (nus_host_pin -> hipHostRegister), pinned memory it allocated itself (hipHostMalloc: the upscaler's slots, the interpolator's
// and the flow estimator's staging, the transfer ring of nus_download / nus_upload) and transparent-huge-page hints it left on
// callers' result buffers (parallel_populate_prepare).  Two fixed-size tables, no allocation, every field an atomic, so that
//   * a fatal-signal handler can print them (nus_fatal_trace.cpp): round 5's GPU fault -- ROCr's "Write access to a read-only
//     page" at a host address -- could only be reasoned about because nobody had kept such a record;
//   * nus_host_unpin can say "this pointer was never pinned here" instead of passing an arbitrary pointer to the runtime.
// Not in the reference (wgpu owns its staging belts: upscale/mod.rs:1010-1057); diagnostics of the drop-in's host side.
#pragma once

#include <cstddef>
#include <cstdint>

namespace nus {

enum RangeKind : uint32_t {
    kRangePinned = 1,     // nus_host_pin: a CALLER's buffer registered with hipHostRegister
    kRangeHostAlloc = 2,  // hipHostMalloc by the library (slots, staging, transfer ring)
    kRangeHugeHint = 3,   // MADV_HUGEPAGE left on a caller's fresh result buffer (history only: it goes with the mapping)
};

struct RangeRecord {
    uint64_t seq;   // order of the event, process-wide, from 1
    uintptr_t lo;   // first byte
    uintptr_t hi;   // one past the last byte
    uint32_t kind;  // RangeKind
    uint32_t op;    // 1 note, 0 forget (history table only; live entries are always 1)
};

// Adds a live entry (and a history line).  False when the live table is full (256 entries): the caller carries on, the
// record is then incomplete and range_overflowed() says so.
bool range_note(RangeKind kind, const void *p, size_t bytes);
// Removes the live entry that starts at `p` (and adds a history line); false when there is none.
bool range_forget(RangeKind kind, const void *p);
// A history line only (hints: nothing to forget).
void range_event(RangeKind kind, const void *p, size_t bytes);
// Is `p` the start of a live entry of this kind?
bool range_is_live(RangeKind kind, const void *p);

// Async-signal-safe readers: copy up to `cap` records, return how many were written.
size_t range_live_snapshot(RangeRecord *out, size_t cap);
size_t range_history_snapshot(RangeRecord *out, size_t cap); // the last <= 128 events, oldest first
bool range_overflowed();

} // namespace nus

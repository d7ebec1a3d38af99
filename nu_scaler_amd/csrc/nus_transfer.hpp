// nus_transfer.hpp -- the library's own road between HBM and a caller's HOST buffer, for callers of the *_device entry points.
//
// The reference's upscale() always ends with host bytes: map the staging buffer, wait, `to_vec` (upscale/mod.rs:1041-1057).
// The device-resident entry points here leave the fetch to the caller, and the obvious fetch -- hipMemcpy / Tensor.cpu() into
// pageable memory -- makes the HIP runtime pin the caller's pages on the fly and keep that registration in a small cache keyed
// by (address, size); a block the allocator freed, trimmed away and handed out again at the same address meets a registration
// whose pages are gone (ROCr: "Write access to a read-only page" at a host address, profiles/r05_gpu_fault_during_pageable_d2h.txt,
// docs/d2h_fault_analysis.md).  download() / upload() never hand a pageable pointer to the runtime: the DMA engines see only a
// ring of pinned chunks this library allocated with hipHostMalloc (kChunks x kChunkBytes per device, allocated on first use, kept
// for the life of the process), and the helper threads of nus_copy.hpp move the bytes between that ring and the caller's buffer
// while the next chunk is on the wire.  A buffer that IS pinned (hipHostMalloc, nus_host_pin) is copied to directly.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace nus {

// Both return a Status (nus_host.hpp) and leave the text in the thread's error slot.  `stream` must belong to the device that
// owns the device pointer; 0 = that device's null stream.
//
// download: device -> host, ordered after everything enqueued on `stream` before the call.  Returns when host_dst holds the bytes.
int download(void *host_dst, const void *d_src, size_t bytes, hipStream_t stream);
// upload: host -> device.  Returns when host_src may be re-used (every byte has been staged); the device bytes are in place for
// work enqueued on `stream` after the call (and for everybody after a synchronisation of that stream).
int upload(void *d_dst, const void *host_src, size_t bytes, hipStream_t stream);

constexpr size_t kTransferChunkBytes = (size_t)8 << 20;
constexpr int kTransferChunks = 4;

} // namespace nus

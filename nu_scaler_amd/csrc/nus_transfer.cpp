// nus_transfer.cpp -- see nus_transfer.hpp.
#include "nus_transfer.hpp"

#include "nus_copy.hpp"
#include "nus_host.hpp"
#include "nus_host_util.hpp"
#include "nus_ranges.hpp"

#include <mutex>

namespace nus {

namespace {

constexpr int kMaxDevices = 64;

// One ring per device: the chunks are pinned memory of that device's context and the events that say "the DMA engine is done
// with chunk k" are recorded on streams of that device.  A transfer holds the ring's mutex from its first chunk to its last, so
// concurrent callers on one device take turns (they share the two DMA engines anyway).
struct Ring {
    std::mutex m;
    uint8_t *stage[kTransferChunks] = {};
    hipEvent_t done[kTransferChunks] = {};
    bool busy[kTransferChunks] = {}; // an engine may still be reading or writing the chunk: wait for done[k] before touching it
    bool ready = false;
};

Ring &ring_of(int device)
{
    static Ring *rings = new Ring[kMaxDevices]; // never destroyed: the runtime may be gone before a static destructor runs
    return rings[device];
}

int fail(int status, const std::string &msg)
{
    set_thread_error(msg);
    return status;
}

int fail_hip(hipError_t e, const char *what)
{
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? kOutOfMemory : kHipError, fmt("HIP error in %s: %s", what, hipGetErrorString(e)));
}

// the calling thread's current device, put back on every way out
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
        if (prev != device) (void)hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int ensure_ring(Ring &r)
{
    if (r.ready) return kOk;
    for (int k = 0; k < kTransferChunks; ++k) {
        if (!r.stage[k]) {
            NUS_HIP(pinned_alloc(reinterpret_cast<void **>(&r.stage[k]), kTransferChunkBytes));
        }
        if (!r.done[k]) NUS_HIP(hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming));
    }
    r.ready = true;
    return kOk;
}

// which device owns `d` -- and is it device memory at all?
int device_of(const void *d, const char *who, int *device)
{
    if (device_count() <= 0) return fail(kNoDevice, fmt("%s: no HIP device available", who));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d) != hipSuccess) {
        (void)hipGetLastError();
        return fail(kInvalidArgument, fmt("%s: the device pointer is not memory the HIP runtime knows", who));
    }
    if (attr.type != hipMemoryTypeDevice) return fail(kInvalidArgument, fmt("%s: the device pointer does not point into device memory", who));
    if (attr.device < 0 || attr.device >= kMaxDevices) return fail(kInvalidArgument, fmt("%s: device index %d out of range", who, attr.device));
    *device = attr.device;
    return kOk;
}

// a "host" pointer that is really device memory would be handed to memcpy: refuse it
bool points_into_device_memory(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // pageable memory: unknown to the runtime
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

int wait_chunk(Ring &r, int k)
{
    if (!r.busy[k]) return kOk;
    NUS_HIP(hipEventSynchronize(r.done[k]));
    r.busy[k] = false;
    return kOk;
}

} // namespace

int download(void *host_dst, const void *d_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return kOk;
    if (!host_dst || !d_src) return fail(kInvalidArgument, "nus_download: null pointer");
    int device = 0;
    int rc = device_of(d_src, "nus_download", &device);
    if (rc != kOk) return rc;
    if (points_into_device_memory(host_dst)) return fail(kInvalidArgument, "nus_download: the host pointer points into device memory");
    DeviceScope scope(device);
    if (is_pinned_host(host_dst)) { // the engines can write it themselves
        NUS_HIP(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
        NUS_HIP(hipStreamSynchronize(stream));
        return kOk;
    }
    Ring &r = ring_of(device);
    std::lock_guard<std::mutex> lk(r.m);
    rc = ensure_ring(r);
    if (rc != kOk) return rc;
    const size_t n = (bytes + kTransferChunkBytes - 1) / kTransferChunkBytes;
    const uint8_t *src = static_cast<const uint8_t *>(d_src);
    uint8_t *dst = static_cast<uint8_t *>(host_dst);
    auto len_of = [&](size_t i) { return i + 1 < n ? kTransferChunkBytes : bytes - i * kTransferChunkBytes; };
    auto issue = [&](size_t i) -> int {
        const int k = (int)(i % kTransferChunks);
        const int w = wait_chunk(r, k); // (an earlier upload's DMA may still be reading it)
        if (w != kOk) return w;
        NUS_HIP(hipMemcpyAsync(r.stage[k], src + i * kTransferChunkBytes, len_of(i), hipMemcpyDeviceToHost, stream));
        NUS_HIP(hipEventRecord(r.done[k], stream));
        r.busy[k] = true;
        return kOk;
    };
    // up to kTransferChunks chunks on the wire; chunk j is copied out by the CPU threads while j+1.. are still arriving
    for (size_t i = 0; i < n && i < (size_t)kTransferChunks; ++i)
        if ((rc = issue(i)) != kOk) return rc;
    for (size_t j = 0; j < n; ++j) {
        const int k = (int)(j % kTransferChunks);
        if ((rc = wait_chunk(r, k)) != kOk) return rc;
        parallel_copy(dst + j * kTransferChunkBytes, r.stage[k], len_of(j));
        if (j + kTransferChunks < n && (rc = issue(j + kTransferChunks)) != kOk) return rc;
    }
    return kOk;
}

int upload(void *d_dst, const void *host_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return kOk;
    if (!d_dst || !host_src) return fail(kInvalidArgument, "nus_upload: null pointer");
    int device = 0;
    int rc = device_of(d_dst, "nus_upload", &device);
    if (rc != kOk) return rc;
    if (points_into_device_memory(host_src)) return fail(kInvalidArgument, "nus_upload: the host pointer points into device memory");
    DeviceScope scope(device);
    if (is_pinned_host(host_src)) {
        NUS_HIP(hipMemcpyAsync(d_dst, host_src, bytes, hipMemcpyHostToDevice, stream));
        NUS_HIP(hipStreamSynchronize(stream)); // the contract: host_src may be re-used on return
        return kOk;
    }
    Ring &r = ring_of(device);
    std::lock_guard<std::mutex> lk(r.m);
    rc = ensure_ring(r);
    if (rc != kOk) return rc;
    const size_t n = (bytes + kTransferChunkBytes - 1) / kTransferChunkBytes;
    const uint8_t *src = static_cast<const uint8_t *>(host_src);
    uint8_t *dst = static_cast<uint8_t *>(d_dst);
    auto len_of = [&](size_t i) { return i + 1 < n ? kTransferChunkBytes : bytes - i * kTransferChunkBytes; };
    // The staging copies of up to kTransferChunks - 1 chunks run on the pool's workers AHEAD of the chunk whose DMA is being issued:
    // one ticket per ring slot; on every way out the tickets are waited for (their pieces point into the caller's buffer).
    struct Tickets {
        CopyTicket t[kTransferChunks];
        ~Tickets()
        {
            for (CopyTicket &x : t) parallel_copy_wait(x);
        }
    } tk;
    auto stage = [&](size_t i) -> int { // chunk i into its slot, asynchronously, once the DMA that last read the slot has finished
        const int k = (int)(i % kTransferChunks);
        const int w = wait_chunk(r, k);
        if (w != kOk) return w;
        parallel_copy_async(r.stage[k], src + i * kTransferChunkBytes, len_of(i), tk.t[k]);
        return kOk;
    };
    for (size_t i = 0; i < n && i + 1 < (size_t)kTransferChunks; ++i)
        if ((rc = stage(i)) != kOk) return rc;
    for (size_t j = 0; j < n; ++j) {
        const int k = (int)(j % kTransferChunks);
        parallel_copy_wait(tk.t[k]); // (the calling thread copies pieces too while it waits)
        NUS_HIP(hipMemcpyAsync(dst + j * kTransferChunkBytes, r.stage[k], len_of(j), hipMemcpyHostToDevice, stream));
        NUS_HIP(hipEventRecord(r.done[k], stream));
        r.busy[k] = true;
        // the slot to stage next is the one of chunk j - 1 ... whose DMA sits in front of the one just issued: waiting for it leaves
        // the engine with work queued
        const size_t nxt = j + kTransferChunks - 1;
        if (nxt < n && (rc = stage(nxt)) != kOk) return rc;
    }
    return kOk;
}

} // namespace nus

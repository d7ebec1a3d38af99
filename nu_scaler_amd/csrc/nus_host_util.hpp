// nus_host_util.hpp -- small helpers shared by the host classes (nus_host.cpp, nus_host_interp.cpp).  Internal.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "nus_ranges.hpp"

namespace nus {

namespace {

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

// True when `p` is host memory the DMA engines can address directly
// (hipHostMalloc / hipHostRegister); pageable memory goes through pinned staging.
bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // pageable memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// hipHostMalloc / hipHostFree that keep the library's record of its own pinned memory (nus_ranges.hpp)
hipError_t pinned_alloc(void **p, size_t bytes)
{
    const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) range_note(kRangeHostAlloc, *p, bytes);
    return e;
}

void pinned_free(void *p)
{
    if (!p) return;
    range_forget(kRangeHostAlloc, p);
    (void)hipHostFree(p);
}

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

} // namespace

} // namespace nus

// inside a member function of a class with fail_hip(): propagate a HIP error as a status code
#define NUS_HIP(call)                                     \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return fail_hip(e_, #call); \
    } while (0)

// probe_pageable_d2h.cpp -- what the HIP runtime does with a device-to-host copy into PAGEABLE memory, observed from outside and
// without provoking anything: every host block used here is its own anonymous mapping that stays mapped until the process ends.
// (docs/d2h_fault_analysis.md: round 5's GPU fault was a write by the runtime's own copy into a brk-heap block.)
//
//   1. which road a copy of `bytes` takes: run under AMD_LOG_LEVEL=4 and grep the log for "Using Pinned resource" /
//      "Using Staging resource" (tools/probe_pageable_d2h.sh does);
//   2. is there a cache of on-the-fly pins keyed by address?  Time copies into ONE block again and again (a cache hit every
//      time after the first) against copies that rotate over 12 blocks (more than a small cache holds: a re-pin every time),
//      and against copies into memory pinned beforehand (no pin work at all).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_pageable_d2h tools/probe_pageable_d2h.cpp
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/mman.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 0) : 33177600; // one 4K RGBA8 frame
    const int reps = argc > 2 ? atoi(argv[2]) : 24;
    const int nblocks = 12;
    void *d = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 0x3C, bytes));
    std::vector<char *> blk(nblocks);
    for (auto &b : blk) {
        b = static_cast<char *>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (b == MAP_FAILED) return 2;
        memset(b, 1, bytes); // resident before any timing
    }
    void *pinned = nullptr;
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    memset(pinned, 1, bytes);
    CK(hipDeviceSynchronize());
    auto run = [&](const char *name, auto pick) -> int {
        std::vector<double> t;
        for (int i = 0; i < reps; ++i) {
            void *dst = pick(i);
            const double t0 = now_ms();
            CK(hipMemcpy(dst, d, bytes, hipMemcpyDeviceToHost));
            t.push_back(now_ms() - t0);
            if (static_cast<unsigned char *>(dst)[bytes - 1] != 0x3C) return 3;
        }
        const double first = t[0];
        std::sort(t.begin() + 1, t.end());
        const double med = t[1 + (t.size() - 1) / 2];
        printf("%-44s first %7.3f ms   median of the rest %7.3f ms  (%6.2f GB/s)\n", name, first, med, bytes / med * 1e-6);
        return 0;
    };
    printf("bytes per copy %zu, %d copies per case\n", bytes, reps);
    if (int r = run("pinned destination (hipHostMalloc)", [&](int) { return pinned; })) return r;
    if (int r = run("pageable, the same block every time", [&](int) { return (void *)blk[0]; })) return r;
    if (int r = run("pageable, rotating over 12 blocks", [&](int i) { return (void *)blk[i % nblocks]; })) return r;
    if (int r = run("pageable, rotating over 4 blocks", [&](int i) { return (void *)blk[i % 4]; })) return r;
    CK(hipDeviceSynchronize());
    return 0;
}

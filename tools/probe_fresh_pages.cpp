// probe_fresh_pages.cpp -- what a fresh 33 MB result buffer costs before a frame can be copied into it (round 4, host path).
// The pyo3-shaped `upscale(bytes) -> bytes` returns a new 33 MB object per call: glibc maps it (above its mmap threshold) and
// unmaps it when the caller drops it, so every result is ~8 100 untouched 4-KiB pages.  Per mode, on fresh mappings: time to make
// the pages present (first touch / MADV_POPULATE_WRITE, with and without MADV_HUGEPAGE, 1 and 4 threads), then the time of a
// memcpy of a resident 33 MB frame into them, and the memcpy straight into untouched pages for comparison.
//   g++ -O2 -std=c++17 -pthread tools/probe_fresh_pages.cpp -o /tmp/probe_fresh_pages && /tmp/probe_fresh_pages
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <sys/mman.h>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void cat(const char *path)
{
    FILE *f = fopen(path, "r");
    char buf[256] = "";
    if (f && fgets(buf, sizeof buf, f)) printf("%s: %s", path, buf);
    else printf("%s: unreadable\n", path);
    if (f) fclose(f);
}

template <typename F>
static void split(char *p, size_t n, int threads, F fn)
{
    if (threads <= 1) return fn(p, n);
    std::vector<std::thread> ts;
    const size_t part = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t) {
        const size_t off = (size_t)t * part;
        if (off >= n) break;
        ts.emplace_back(fn, p + off, std::min(part, n - off));
    }
    for (auto &t : ts) t.join();
}

int main()
{
    cat("/sys/kernel/mm/transparent_hugepage/enabled");
    cat("/sys/kernel/mm/transparent_hugepage/defrag");
    cat("/sys/kernel/mm/transparent_hugepage/hpage_pmd_size");
    const size_t n = (size_t)3840 * 2160 * 4;
    std::vector<char> src(n, 7);
    struct Mode { const char *name; bool huge; int how; }; // how: 0 touch, 1 populate, 2 nothing (memcpy takes the faults)
    const Mode modes[] = {{"first touch (1 byte per page)", false, 0}, {"MADV_POPULATE_WRITE", false, 1},
                          {"MADV_HUGEPAGE + first touch", true, 0}, {"MADV_HUGEPAGE + MADV_POPULATE_WRITE", true, 1},
                          {"nothing (memcpy faults)", false, 2}, {"MADV_HUGEPAGE, nothing (memcpy faults)", true, 2}};
    for (int threads : {1, 4}) {
        printf("-- %d thread(s): ms to make 33.2 MB present | ms of the memcpy into it afterwards | sum (median of 7 fresh mappings)\n", threads);
        for (const Mode &m : modes) {
            std::vector<double> a, b;
            for (int rep = 0; rep < 7; ++rep) {
                char *p = static_cast<char *>(mmap(nullptr, n + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
                if (p == MAP_FAILED) return 1;
                if (m.huge) madvise(p, n + 4096, MADV_HUGEPAGE);
                char *q = p + 32; // (a malloc'd chunk starts behind its header)
                double t0 = now_ms();
                if (m.how == 0)
                    split(q, n, threads, [](char *s, size_t len) { for (size_t i = 0; i < len; i += 4096) __atomic_fetch_add(s + i, 0, __ATOMIC_RELAXED); });
                else if (m.how == 1)
                    split(q, n, threads, [](char *s, size_t len) {
                        const uintptr_t lo = ((uintptr_t)s + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)s + len) & ~(uintptr_t)4095;
                        if (hi > lo && madvise((void *)lo, hi - lo, MADV_POPULATE_WRITE) != 0) perror("madvise");
                    });
                double t1 = now_ms();
                const char *s0 = src.data();
                split(q, n, threads, [q, s0](char *d, size_t len) { memcpy(d, s0 + (d - q), len); });
                double t2 = now_ms();
                a.push_back(t1 - t0);
                b.push_back(t2 - t1);
                munmap(p, n + 4096);
            }
            std::sort(a.begin(), a.end());
            std::sort(b.begin(), b.end());
            printf("%-44s %7.3f | %7.3f | %7.3f\n", m.name, a[3], b[3], a[3] + b[3]);
        }
    }
    return 0;
}

// Host-only stress test of nu_scaler_amd/csrc/nus_copy.cpp (built and run by tests/test_host_logic.py):
// copies of random sizes and alignments from several threads at once, every result compared with the source;
// then the same from a forked child.
#include "nus_copy.hpp"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

static uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main()
{
    std::atomic<int> bad{0};
    auto work = [&](int id) {
        uint64_t seed = 0x5EED + id;
        const size_t cap = (size_t)24 << 20;
        std::vector<uint8_t> src(cap + 64), dst(cap + 64);
        for (size_t i = 0; i < src.size(); i += 8) {
            const uint64_t v = splitmix(seed);
            memcpy(&src[i], &v, src.size() - i < 8 ? src.size() - i : 8);
        }
        for (int round = 0; round < 40; ++round) {
            const size_t n = round < 4 ? (size_t[]){0, 1, (1u << 20) - 1, 1u << 20}[round] : (size_t)(splitmix(seed) % cap);
            const size_t so = splitmix(seed) % 64, dof = splitmix(seed) % 64;
            memset(dst.data(), 0xA5, dst.size());
            nus::parallel_copy(dst.data() + dof, src.data() + so, n);
            if (memcmp(dst.data() + dof, src.data() + so, n) != 0) ++bad;
            for (size_t i = 0; i < dof; ++i) bad += dst[i] != 0xA5;             // nothing before
            for (size_t i = dof + n; i < dof + n + 32 && i < dst.size(); ++i) bad += dst[i] != 0xA5; // nothing after
        }
    };
    std::vector<std::thread> threads;
    for (int t = 0; t < 4; ++t) threads.emplace_back(work, t);
    for (auto &t : threads) t.join();
    // populate requests (parallel_populate_async: the pages of a result buffer made present while its frame is on the GPU)
    // racing with copies into the SAME fresh anonymous mapping: no byte of the copy may be lost or changed, untouched
    // bytes stay zero, and the ticket drains
    for (int round = 0; round < 12; ++round) {
        uint64_t seed = 0xB00 + round;
        const size_t n = ((size_t)6 << 20) + (splitmix(seed) % ((size_t)20 << 20)), off = splitmix(seed) % 4096;
        char *fresh = static_cast<char *>(mmap(nullptr, n + 8192, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (fresh == MAP_FAILED) { ++bad; break; }
        std::vector<uint8_t> src(n);
        for (size_t i = 0; i + 8 <= n; i += 8) {
            const uint64_t v = splitmix(seed) | 1;
            memcpy(&src[i], &v, 8);
        }
        nus::CopyTicket pop, cp;
        if (nus::parallel_copy_workers() > 0 && !nus::parallel_populate_prepare(fresh + off, n)) ++bad; // fresh: wants requests
        nus::parallel_populate_async(fresh + off, n, pop);
        const size_t half = n / 2;
        nus::parallel_copy_async(fresh + off, src.data(), half, cp);       // first half while the populate pieces are queued
        nus::parallel_copy_wait(cp);
        nus::parallel_populate_async(fresh + off, n, pop);                  // again, now partly resident
        nus::parallel_copy_async(fresh + off + half, src.data() + half, n - half - 4096, cp); // the last 4 KiB stay untouched
        nus::parallel_copy_wait(cp);
        nus::parallel_copy_wait(pop);
        if (pop.left != 0 || memcmp(fresh + off, src.data(), n - 4096) != 0) ++bad;
        if (nus::parallel_populate_prepare(fresh + off, n)) ++bad; // resident now: nothing to do
        for (size_t i = n - 4096; i < n; ++i) bad += fresh[off + i] != 0;
        munmap(fresh, n + 8192);
    }
    // a forked child has the pool's state but none of its threads: it must still copy (alone) and exit cleanly
    const pid_t pid = fork();
    if (pid == 0) {
        std::vector<uint8_t> a((size_t)5 << 20, 7), b((size_t)5 << 20, 0);
        nus::parallel_copy(b.data(), a.data(), a.size());
        return memcmp(a.data(), b.data(), a.size()) == 0 ? 0 : 3;
    }
    int status = 0;
    if (pid < 0 || waitpid(pid, &status, 0) != pid || !WIFEXITED(status) || WEXITSTATUS(status) != 0) ++bad;
    printf("workers %d bad %d\n", nus::parallel_copy_workers(), bad.load());
    return bad.load() == 0 ? 0 : 1;
}

// Host-only stress test of nu_scaler_amd/csrc/nus_copy.cpp (built and run by tests/test_host_logic.py):
// copies of random sizes and alignments from several threads at once, every result compared with the source;
// then the same from a forked child.
#include "nus_copy.hpp"
#include "nus_ranges.hpp"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <cstdlib>
#include <malloc.h>
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

#define BAD() (fprintf(stderr, "check failed at line %d\n", __LINE__), ++bad)
#define NOTE(msg) fprintf(stderr, "%s (line %d)\n", msg, __LINE__)

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
    // populate requests on every kind of caller memory the host path can be handed (round 5, after the unexplained abort of
    // round 4): blocks shorter than a page and blocks that straddle one page boundary (nothing to do, nothing touched), a piece
    // of the program break's heap and a piece of a larger malloc block (populated, never hinted), a block that is its own
    // mapping (hinted), and every one of them given back to the allocator IMMEDIATELY after the wait -- a request that outlived
    // its wait would then touch freed or unmapped memory (the sanitizer builds of this program report it)
    {
        // never a page of its own: the predicate that guards the transparent-huge-page hint
        std::vector<char> small(1 << 16, 1);
        if (nus::parallel_populate_own_mapping(small.data(), small.size())) BAD();
        if (!nus::parallel_populate_own_mapping(small.data(), (size_t)32 << 20)) BAD(); // by size alone (never dereferenced)
        for (size_t len : {(size_t)1, (size_t)100, (size_t)4095, (size_t)4097, (size_t)8191}) {
            char *b = static_cast<char *>(malloc(len + 4096));
            memset(b, 0x5A, len + 4096);
            nus::CopyTicket t;
            nus::parallel_populate_async(b + 7, len, t);
            nus::parallel_copy_wait(t);
            for (size_t i = 0; i < len + 4096; ++i) bad += b[i] != 0x5A;
            free(b);
        }
#if !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__)
        // the program break's heap: with the mmap threshold at its maximum (what glibc's dynamic threshold reaches by itself once
        // a 32 MB block has been freed) a 9 MB block is carved out of it, and with a small trim threshold free() gives the
        // range back to the kernel at once (the sanitizers' allocators do neither)
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        mallopt(M_TRIM_THRESHOLD, 128 << 10);
        for (int round = 0; round < 3; ++round) {
            const size_t brk_len = (size_t)9 << 20;
            char *base = static_cast<char *>(malloc(brk_len));
            const bool on_brk = base < static_cast<char *>(sbrk(0)) && base + brk_len <= static_cast<char *>(sbrk(0));
            if (!on_brk) NOTE("(malloc did not use the program break: skipped)");
            if (on_brk && nus::parallel_populate_own_mapping(base + 32, brk_len - 32)) BAD(); // a heap chunk, not a mapping
            nus::CopyTicket t;
            (void)nus::parallel_populate_prepare(base + 32, brk_len - 32);
            nus::parallel_populate_async(base + 32, brk_len - 32, t);
            nus::parallel_copy_wait(t);
            free(base); // top of the heap, above the trim threshold: unmapped now
        }
        mallopt(M_MMAP_THRESHOLD, 128 << 10); // back to the default for the blocks below
        // blocks the allocator maps for themselves: glibc's mmapped chunks (33 MB is above the initial threshold of 128 KiB; both
        // are allocated before either is freed -- freeing one raises the threshold to its size and the next 33 MB block comes
        // from the program break, which is exactly the case the predicate must answer "no" to, checked last)
        {
            const size_t n = (size_t)3840 * 2160 * 4, offs[2] = {0, 32}; // a Vec's pointer; a PyBytes' (32-byte object header first)
            char *blk[2] = {static_cast<char *>(malloc(n + offs[0])), static_cast<char *>(malloc(n + offs[1]))};
            for (int k = 0; k < 2; ++k) {
                const size_t off = offs[k];
                if (!nus::parallel_populate_own_mapping(blk[k] + off, n)) BAD();
                if (nus::parallel_populate_own_mapping(blk[k] + off + 8192, n - 8192)) BAD(); // the inside of a block is not a block
                nus::CopyTicket t;
                if (nus::parallel_copy_workers() > 0 && !nus::parallel_populate_prepare(blk[k] + off, n)) BAD();
                nus::parallel_populate_async(blk[k] + off, n, t);
                nus::parallel_copy_wait(t);
            }
            free(blk[0]); // munmap: a late request would fault
            free(blk[1]);
            char *again = static_cast<char *>(malloc(n)); // now served from the heap (dynamic mmap threshold): never hinted
            const bool heap_block = (reinterpret_cast<uintptr_t>(again) & 4095) != 16;
            if (heap_block && nus::parallel_populate_own_mapping(again, n)) BAD();
            nus::CopyTicket t;
            nus::parallel_populate_async(again, n, t);
            nus::parallel_copy_wait(t);
            free(again);
        }
        // a piece of a larger malloc block: its own pages, but not its own mapping
        {
            const size_t n = (size_t)40 << 20;
            char *blk = static_cast<char *>(calloc(1, n));
            if (nus::parallel_populate_own_mapping(blk + ((size_t)4 << 20) + 16, (size_t)20 << 20)) BAD();
            free(blk);
        }
#endif
        // many tickets in flight at once, each buffer freed right after its own wait while the others' requests are still queued
        std::vector<std::thread> th;
        for (int k = 0; k < 4; ++k)
            th.emplace_back([&, k] {
                for (int round = 0; round < 16; ++round) {
                    const size_t n = ((size_t)2 << 20) + (size_t)((k * 16 + round) * 123457 % (6 << 20));
                    char *b = static_cast<char *>(malloc(n));
                    std::vector<char> src(n, (char)(k + round));
                    nus::CopyTicket pop, cp;
                    nus::parallel_populate_async(b, n, pop);
                    nus::parallel_copy_async(b, src.data(), n, cp);
                    nus::parallel_copy_wait(cp);
                    nus::parallel_copy_wait(pop);
                    if (pop.left != 0 || cp.left != 0 || memcmp(b, src.data(), n) != 0) BAD();
                    free(b);
                }
            });
        for (auto &t : th) t.join();
        if (nus::parallel_copy_pending() != 0) BAD(); // nothing may outlive the waits
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
    // the record of host ranges (nus_ranges.hpp): four threads note and forget at once; afterwards exactly the entries that were
    // not forgotten are live, each found by its start address, and the history holds the last events in order
    {
        auto ranges = [&](int id) {
            for (uintptr_t i = 0; i < 200; ++i) {
                void *p = reinterpret_cast<void *>(((uintptr_t)(id + 1) << 32) + (i << 12));
                if (!nus::range_note(nus::kRangePinned, p, 4096)) BAD();
                if (!nus::range_is_live(nus::kRangePinned, p)) BAD();
                if (nus::range_is_live(nus::kRangeHostAlloc, p)) BAD(); // another kind at the same address is another entry
                if (i % 8 != 0 && !nus::range_forget(nus::kRangePinned, p)) BAD();
                if (i % 8 != 0 && nus::range_forget(nus::kRangePinned, p)) BAD(); // twice: the second finds nothing
            }
        };
        std::vector<std::thread> rt;
        for (int t = 0; t < 4; ++t) rt.emplace_back(ranges, t);
        for (auto &t : rt) t.join();
        static nus::RangeRecord recs[256];
        const size_t live = nus::range_live_snapshot(recs, 256);
        if (live != 4 * 25 || nus::range_overflowed()) BAD();
        for (size_t i = 0; i < live; ++i)
            if (recs[i].kind != nus::kRangePinned || recs[i].hi - recs[i].lo != 4096 || ((recs[i].lo >> 12) & 7) != 0) BAD();
        const size_t hist = nus::range_history_snapshot(recs, 256);
        if (hist == 0 || hist > 128) BAD();
        for (size_t i = 1; i < hist; ++i)
            if (recs[i].seq == recs[i - 1].seq) BAD();
        size_t taken = live; // fill the table (256 slots): one note more is refused, and the record says it is incomplete
        for (uintptr_t i = 0; taken < 256; ++i, ++taken)
            if (!nus::range_note(nus::kRangeHostAlloc, reinterpret_cast<void *>((uintptr_t)9 << 40 | i << 12), 4096)) BAD();
        if (nus::range_note(nus::kRangeHostAlloc, reinterpret_cast<void *>((uintptr_t)10 << 40), 4096) || !nus::range_overflowed()) BAD();
    }
    printf("workers %d bad %d\n", nus::parallel_copy_workers(), bad.load());
    return bad.load() == 0 ? 0 : 1;
}

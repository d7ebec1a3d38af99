// nus_copy.cpp -- see nus_copy.hpp.
#include "nus_copy.hpp"

#include "nus_ranges.hpp"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>

#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace nus {

namespace {

constexpr size_t kMinParallelBytes = 1u << 20; // below this one memcpy is faster than waking anybody
constexpr size_t kPieceBytes = 512u << 10;     // work item: large enough to stream, small enough to balance
constexpr size_t kPopulateBytes = 2u << 20;    // work item of a populate request

// Make the pages of [p, p + len) present and writable without changing a byte of them.  A fresh 33 MB result buffer (the
// Vec / PyBytes the trait's `upscale` returns: an anonymous mapping the allocator has just been given, and gives back when
// the caller drops the result) costs ~8 100 first-touch faults; taken inside the copy-out they sit in the frame's critical
// path (measured through the pyo3 shapes: 1.80 instead of 0.98 ms per call, 2.9 ms per frame of a batch), taken here they run
// beside the frame's DMA and kernel.  MADV_POPULATE_WRITE (Linux 5.14) where the kernel has it, else an atomic add of zero to
// one byte per page -- atomic, so a copy piece that writes the same page at the same time loses nothing.
// MADV_POPULATE_WRITE is asked about ONCE, on a page this library maps for itself: EINVAL on a caller's range also means "this
// mapping cannot be populated" (VM_PFNMAP / VM_IO, no write permission), and must not turn the call off for the whole process.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
bool kernel_has_populate_write()
{
    static const bool have = [] {
        void *pg = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (pg == MAP_FAILED) return false;
        const bool ok = madvise(pg, 4096, MADV_POPULATE_WRITE) == 0;
        munmap(pg, 4096);
        return ok;
    }();
    return have;
}

void populate_pages(char *p, size_t len)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uintptr_t lo = (a + 4095) & ~(uintptr_t)4095, hi = (a + len) & ~(uintptr_t)4095;
    if (hi <= lo) return;
    if (kernel_has_populate_write()) {
        // Any failure (ENOMEM: part of the range is not mapped; EFAULT, EINVAL: a mapping that cannot be populated; EAGAIN) leaves
        // the pages to the copy-out's own first-touch faults: this is an optimisation, and a range the kernel refuses is not one
        // to dereference.
        (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_WRITE);
        return;
    }
    // a kernel older than 5.14 (probed above): touch the pages instead.  Only ranges a caller has handed over as writable output
    // get here, exactly as the copy-out would write them a moment later.
    for (uintptr_t q = lo; q < hi; q += 4096) (void)__atomic_fetch_add(reinterpret_cast<volatile char *>(q), 0, __ATOMIC_RELAXED);
}

// Is [p, p + len) a block that is its OWN mapping -- one the allocator maps for this block alone and unmaps when the block is
// freed, so that a hint left on it (MADV_HUGEPAGE) goes away with it?  True for blocks of at least 32 MiB (glibc's mmap threshold
// never grows past that: DEFAULT_MMAP_THRESHOLD_MAX) and for blocks that carry glibc's mmapped-chunk header: the word in front
// of the first user byte of the mapping holds the chunk size (a whole number of pages, just enough for this block) with
// IS_MMAPPED (2) set and PREV_INUSE (1) / NON_MAIN_ARENA (4) clear.  The user pointer may sit a few bytes into the chunk (a PyBytes
// keeps a 32-byte object header in front of its data; a Vec / numpy array none): the header is looked for at the start of the
// pointer's own page, which is readable because the block's first byte is.  Any other allocator, and blocks glibc carved out of
// a heap (the program break's, or a thread arena's 64-MiB regions), answer false: they keep their address range when the block
// is freed and a hint would stay on a piece of the process heap for good.
bool is_own_mapping(const void *p, size_t len)
{
    if (len >= ((size_t)32 << 20)) return true;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p), page = a & ~(uintptr_t)4095, off = a - page;
    if (off < 16 || off > 256) return false; // mmapped chunk: 16 bytes of chunk header at the start of the mapping, then the block
    size_t hdr[2];
    memcpy(hdr, reinterpret_cast<const void *>(page), sizeof(hdr));
    const size_t size = hdr[1] & ~(size_t)7;
    if (hdr[0] != 0 || (hdr[1] & 7) != 2 || size % 4096 != 0) return false;
    return size >= off + len && size <= off + len + 2 * 4096;
}

// A queue of pieces (dst, src, len, ticket; src == nullptr: populate dst's pages) shared by every caller: several copies can be in flight at once -- the staging
// copy of frame i+1 and the copy-out of frame i of upscale_batch -- and whoever has a hand free takes the next piece.
struct Piece {
    char *dst;
    const char *src;
    size_t len;
    CopyTicket *ticket;
};

class CopyPool {
public:
    // Never destroyed: a forked child inherits the mutex / condition variables with the parent's waiters recorded in
    // them, and destroying those would block for ever.  The owning process stops and joins its workers in shutdown().
    static CopyPool &instance()
    {
        static CopyPool *pool = new CopyPool;
        return *pool;
    }
    static void shutdown_at_unload() { instance().shutdown(); }

    int workers() const { return getpid() == owner_ ? (int)threads_.size() : 0; }

    size_t pending()
    {
        std::lock_guard<std::mutex> lk(m_);
        return queue_.size() + queue_lo_.size() + running_;
    }

    // Queue the pieces of one copy under `ticket` and wake the workers.  The bytes are copied by the time ticket.wait()
    // (or help()) returns; the caller keeps both buffers alive until then.
    void submit(char *dst, const char *src, size_t bytes, CopyTicket &ticket)
    {
        const size_t pieces = (bytes + kPieceBytes - 1) / kPieceBytes;
        if (pieces == 0) return;
        {
            std::lock_guard<std::mutex> lk(m_);
            ticket.left += pieces;
            for (size_t i = 0; i < pieces; ++i) {
                const size_t off = i * kPieceBytes;
                queue_.push_back(Piece{dst + off, src + off, bytes - off < kPieceBytes ? bytes - off : kPieceBytes, &ticket});
            }
        }
        cv_work_.notify_all();
    }

    // Queue "make these pages present" requests (see populate_pages) under `ticket`: pieces cut at 2-MiB ADDRESS boundaries, so
    // that with transparent huge pages each request faults whole huge pages in.
    void submit_populate(char *dst, size_t bytes, CopyTicket &ticket)
    {
        if (bytes == 0) return;
        std::vector<Piece> pieces;
        const uintptr_t a = reinterpret_cast<uintptr_t>(dst), end = a + bytes;
        for (uintptr_t lo = a; lo < end;) {
            const uintptr_t hi = std::min<uintptr_t>((lo / kPopulateBytes + 1) * kPopulateBytes, end);
            pieces.push_back(Piece{reinterpret_cast<char *>(lo), nullptr, (size_t)(hi - lo), &ticket});
            lo = hi;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            ticket.left += pieces.size();
            ticket.low = true;
            for (const Piece &p : pieces) queue_lo_.push_back(p); // behind every copy: a copy is somebody's critical path
        }
        cv_work_.notify_all();
    }

    // Work on queued pieces (anybody's) until `ticket` has none left: the calling thread is one more worker, and alone
    // finishes the job where no worker exists (NUS_COPY_THREADS=0, a forked child).
    void help(CopyTicket &ticket)
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (ticket.left == 0) return;
            // (a thread waiting for a copy does not pick up populate requests: its copy is on a frame's critical path)
            if (queue_.empty() && (!ticket.low || queue_lo_.empty())) { // the rest of this ticket is in other threads' hands
                cv_done_.wait(lk, [&] { return ticket.left == 0 || !queue_.empty() || (ticket.low && !queue_lo_.empty()); });
                continue;
            }
            run_one(lk);
        }
    }

private:
    CopyPool()
    {
        // Default: 6 workers where the process has the CPUs for them next to the submitting and the retiring thread of the
        // host path -- its affinity mask, capped by the cgroup's CPU quota (a GPU box of this pool gives a job 16 CPUs' worth of
        // a 256-thread host).  Three were enough while the pool only copied; since it also makes fresh result buffers present
        // (populate_pages) a batch of fresh outputs goes 1.05 -> 0.72 ms per frame with six (profiles/r04_host_path_pool_threads.txt).
        int n = 6;
        const bool from_env = getenv("NUS_COPY_THREADS") != nullptr;
        if (from_env) n = atoi(getenv("NUS_COPY_THREADS"));
        int cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (int)std::thread::hardware_concurrency();
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
                const int q = (int)((quota + period / 2) / period);
                if (q >= 1 && (cpus <= 0 || q < cpus)) cpus = q;
            }
            fclose(f);
        }
        if (cpus > 0 && !from_env && n + 2 > cpus) n = cpus - 2; // leave the two threads of the host path their CPUs
        if (cpus > 0 && (n < 0 ? 0 : n) + 1 > cpus) n = cpus - 1;
        if (n > 8) n = 8;
        if (n < 0) n = 0;
        owner_ = getpid();
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { worker(); });
        if (n > 0) atexit(&CopyPool::shutdown_at_unload); // library unload / process exit: no thread may outlive the code
    }

    void shutdown()
    {
        if (getpid() != owner_) return; // the workers did not survive a fork: nothing to stop
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread &t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
    }

    // pops one piece and copies it with the lock released; called and returns with `lk` held
    void run_one(std::unique_lock<std::mutex> &lk)
    {
        std::deque<Piece> &q = queue_.empty() ? queue_lo_ : queue_;
        const Piece p = q.front();
        q.pop_front();
        ++running_;
        lk.unlock();
        if (p.src)
            memcpy(p.dst, p.src, p.len);
        else
            populate_pages(p.dst, p.len);
        lk.lock();
        --running_;
        if (--p.ticket->left == 0) cv_done_.notify_all();
    }

    void worker()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || !queue_.empty() || !queue_lo_.empty(); });
            if (stop_) return;
            run_one(lk);
        }
    }

    std::mutex m_; // guards the queue, every ticket's count, stop_
    std::condition_variable cv_work_, cv_done_;
    std::deque<Piece> queue_;    // copies
    std::deque<Piece> queue_lo_; // populate requests: taken when no copy is waiting
    std::vector<std::thread> threads_;
    size_t running_ = 0; // pieces popped and not yet finished
    bool stop_ = false;
    pid_t owner_ = 0;
};

} // namespace

void parallel_copy_async(void *dst, const void *src, size_t bytes, CopyTicket &ticket)
{
    CopyPool::instance().submit(static_cast<char *>(dst), static_cast<const char *>(src), bytes, ticket);
}

void parallel_copy_wait(CopyTicket &ticket) { CopyPool::instance().help(ticket); }

bool parallel_populate_prepare(void *dst, size_t bytes)
{
    if (bytes < kMinParallelBytes) return false; // a small buffer's few faults are cheaper than a wake-up
    if (CopyPool::instance().workers() == 0) return false; // nobody to run beside the frame: the copy-out takes the faults as before
    const uintptr_t a = reinterpret_cast<uintptr_t>(dst);
    const uintptr_t lo = (a + 4095) & ~(uintptr_t)4095, hi = (a + bytes) & ~(uintptr_t)4095;
    if (hi <= lo) return false;
    // Resident already (a buffer the caller re-uses)?  One page per 2 MiB and the last one, asked of mincore (shared lock, a
    // fraction of a microsecond each): then there is nothing to do, and nothing may be done -- see below.
    bool resident = true;
    for (uintptr_t q = lo; resident && q < hi; q = (q / kPopulateBytes + 1) * kPopulateBytes) {
        unsigned char vec = 0;
        resident = mincore(reinterpret_cast<void *>(q), 4096, &vec) == 0 && (vec & 1);
    }
    if (resident) {
        unsigned char vec = 0;
        resident = mincore(reinterpret_cast<void *>(hi - 4096), 4096, &vec) == 0 && (vec & 1);
    }
    if (resident) return false;
    // A fresh mapping: ask for transparent huge pages on its whole pages (a hint on the mapping; a no-op where THP is off) -- 16
    // huge-page faults instead of 8 100 small ones for a 4K frame, and a copy that misses the TLB 512 times less often
    // (33 MB made present and copied into by 4 threads on a GPU box: 1.95 ms through first-touch faults in the copy, 1.46 with
    // MADV_POPULATE_WRITE first, 0.82 with the hint as well: profiles/r04_host_fresh_result_pages.txt).  One call for the whole
    // range, and only for fresh mappings: madvise takes the address space's lock exclusively, i.e. waits for every populate
    // request in flight -- issued per frame on resident buffers it serialised a batch (0.75 -> 1.48 ms per frame).
    // Only on blocks that are their own mapping (is_own_mapping): there the hint is unmapped with the block.  On a block the
    // allocator carved out of a heap -- glibc serves a 33 MB block from the program break once its dynamic mmap threshold has
    // grown past that size, or from a thread arena -- the address range outlives the block and the hint would stay on a piece of
    // the process heap for good (khugepaged would keep working on it): such blocks are populated without it.
    // NUS_NO_THP_HINT (any value): never give the hint -- for processes under an allocator other than glibc's whose blocks of less than
    // 32 MiB might look like glibc's mmapped chunks (is_own_mapping reads glibc's chunk header), or that manage huge pages themselves
    static const bool hint_allowed = getenv("NUS_NO_THP_HINT") == nullptr;
    if (hint_allowed && is_own_mapping(dst, bytes)) {
        (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
        range_event(kRangeHugeHint, reinterpret_cast<void *>(lo), hi - lo); // (nus_ranges.hpp: what a fatal-signal report lists)
    }
    return true;
}

void parallel_populate_async(void *dst, size_t bytes, CopyTicket &ticket)
{
    if (bytes == 0) return;
    CopyPool &pool = CopyPool::instance();
    if (pool.workers() == 0) return;
    pool.submit_populate(static_cast<char *>(dst), bytes, ticket);
}

void parallel_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes < kMinParallelBytes) {
        memcpy(dst, src, bytes);
        return;
    }
    CopyTicket t;
    parallel_copy_async(dst, src, bytes, t);
    parallel_copy_wait(t);
}

int parallel_copy_workers() { return CopyPool::instance().workers(); }

size_t parallel_copy_pending() { return CopyPool::instance().pending(); }

bool parallel_populate_own_mapping(const void *p, size_t bytes) { return is_own_mapping(p, bytes); }

} // namespace nus

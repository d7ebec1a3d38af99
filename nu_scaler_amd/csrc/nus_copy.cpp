// nus_copy.cpp -- see nus_copy.hpp.
#include "nus_copy.hpp"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <atomic>
#include <cerrno>
#include <cstdint>

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
void populate_pages(char *p, size_t len)
{
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uintptr_t lo = (a + 4095) & ~(uintptr_t)4095, hi = (a + len) & ~(uintptr_t)4095;
    if (hi <= lo) return;
    static std::atomic<int> have_madvise{1};
    if (have_madvise.load(std::memory_order_relaxed)) {
        if (madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_WRITE) == 0) return;
        if (errno == EINVAL) have_madvise.store(0, std::memory_order_relaxed); // an older kernel: touch the pages instead
    }
    for (uintptr_t q = lo; q < hi; q += 4096) (void)__atomic_fetch_add(reinterpret_cast<volatile char *>(q), 0, __ATOMIC_RELAXED);
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

    // Queue "make these pages present" requests (see populate_pages) under `ticket`.
    void submit_populate(char *dst, size_t bytes, CopyTicket &ticket)
    {
        const size_t pieces = (bytes + kPopulateBytes - 1) / kPopulateBytes;
        if (pieces == 0) return;
        {
            std::lock_guard<std::mutex> lk(m_);
            ticket.left += pieces;
            for (size_t i = 0; i < pieces; ++i) {
                const size_t off = i * kPopulateBytes;
                queue_.push_back(Piece{dst + off, nullptr, bytes - off < kPopulateBytes ? bytes - off : kPopulateBytes, &ticket});
            }
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
            if (queue_.empty()) { // the rest of this ticket is in other threads' hands
                cv_done_.wait(lk, [&] { return ticket.left == 0 || !queue_.empty(); });
                continue;
            }
            run_one(lk);
        }
    }

private:
    CopyPool()
    {
        int n = 3;
        if (const char *e = getenv("NUS_COPY_THREADS")) n = atoi(e);
        int cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (int)std::thread::hardware_concurrency();
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
        const Piece p = queue_.front();
        queue_.pop_front();
        lk.unlock();
        if (p.src)
            memcpy(p.dst, p.src, p.len);
        else
            populate_pages(p.dst, p.len);
        lk.lock();
        if (--p.ticket->left == 0) cv_done_.notify_all();
    }

    void worker()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (stop_) return;
            run_one(lk);
        }
    }

    std::mutex m_; // guards the queue, every ticket's count, stop_
    std::condition_variable cv_work_, cv_done_;
    std::deque<Piece> queue_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
    pid_t owner_ = 0;
};

} // namespace

void parallel_copy_async(void *dst, const void *src, size_t bytes, CopyTicket &ticket)
{
    CopyPool::instance().submit(static_cast<char *>(dst), static_cast<const char *>(src), bytes, ticket);
}

void parallel_copy_wait(CopyTicket &ticket) { CopyPool::instance().help(ticket); }

void parallel_populate_async(void *dst, size_t bytes, CopyTicket &ticket)
{
    if (bytes < kMinParallelBytes) return; // a small buffer's few faults are cheaper than a wake-up
    CopyPool &pool = CopyPool::instance();
    if (pool.workers() == 0) return; // nobody to run beside the frame: the copy-out takes the faults as before
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

} // namespace nus

// nus_copy.cpp -- see nus_copy.hpp.
#include "nus_copy.hpp"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace nus {

namespace {

constexpr size_t kMinParallelBytes = 1u << 20; // below this one memcpy is faster than waking anybody
constexpr size_t kPieceBytes = 512u << 10;     // work item: large enough to stream, small enough to balance

// One copy at a time.  The job lives on the caller's stack: workers enter it only under the pool mutex while it is
// published, and the caller leaves only when every piece is done AND every worker that entered has left.
struct Job {
    char *dst;
    const char *src;
    size_t bytes, pieces;
    std::atomic<size_t> next{0}, left{0};
    int inside = 0; // workers currently in run(); guarded by the pool mutex
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

    int workers() const { return (int)threads_.size(); }

    void copy(char *dst, const char *src, size_t bytes)
    {
        std::unique_lock<std::mutex> api(api_, std::try_to_lock);
        if (!api.owns_lock() || threads_.empty()) { // pool busy with another caller's copy (or disabled)
            memcpy(dst, src, bytes);
            return;
        }
        Job job;
        job.dst = dst;
        job.src = src;
        job.bytes = bytes;
        job.pieces = (bytes + kPieceBytes - 1) / kPieceBytes;
        job.left.store(job.pieces, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            ++generation_;
        }
        cv_work_.notify_all();
        run(job); // the caller works too, and alone finishes the job if no worker ever wakes (e.g. after fork)
        std::unique_lock<std::mutex> lk(m_);
        job_ = nullptr; // no new entrants
        cv_done_.wait(lk, [&] { return job.left.load(std::memory_order_acquire) == 0 && job.inside == 0; });
    }

private:
    CopyPool()
    {
        int n = 3;
        if (const char *e = getenv("NUS_COPY_THREADS")) n = atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw > 0 && (unsigned)(n < 0 ? 0 : n) + 1 > hw) n = (int)hw - 1;
        if (n > 8) n = 8;
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

    void run(Job &job)
    {
        for (;;) {
            const size_t i = job.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= job.pieces) return;
            const size_t off = i * kPieceBytes;
            memcpy(job.dst + off, job.src + off, job.bytes - off < kPieceBytes ? job.bytes - off : kPieceBytes);
            if (job.left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_); // pairs with the waiter's predicate check
                cv_done_.notify_all();
            }
        }
    }

    void worker()
    {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            Job *job = job_;
            if (job == nullptr) continue; // already finished by the others
            ++job->inside;
            lk.unlock();
            run(*job);
            lk.lock();
            if (--job->inside == 0) cv_done_.notify_all();
        }
    }

    std::mutex api_; // one parallel copy at a time
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> threads_;
    unsigned long long generation_ = 0;
    bool stop_ = false;
    Job *job_ = nullptr;
    pid_t owner_ = 0;
};

} // namespace

void parallel_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes < kMinParallelBytes) {
        memcpy(dst, src, bytes);
        return;
    }
    CopyPool::instance().copy(static_cast<char *>(dst), static_cast<const char *>(src), bytes);
}

int parallel_copy_workers() { return CopyPool::instance().workers(); }

} // namespace nus

// nus_ranges.cpp -- see nus_ranges.hpp.
#include "nus_ranges.hpp"

#include <atomic>

namespace nus {

namespace {

constexpr size_t kLive = 256;
constexpr size_t kHistory = 128;

struct LiveSlot {
    std::atomic<uint32_t> state{0}; // 0 free, 1 being written, 2 valid
    std::atomic<uint32_t> kind{0};
    std::atomic<uint64_t> seq{0};
    std::atomic<uintptr_t> lo{0}, hi{0};
};

struct HistorySlot {
    std::atomic<uint64_t> seq{0}; // written last: 0 = never used / being rewritten
    std::atomic<uintptr_t> lo{0}, hi{0};
    std::atomic<uint32_t> kind{0}, op{0};
};

LiveSlot g_live[kLive];
HistorySlot g_hist[kHistory];
std::atomic<uint64_t> g_seq{0};
std::atomic<uint64_t> g_hist_head{0};
std::atomic<bool> g_overflow{false};

uint64_t history_line(RangeKind kind, uint32_t op, uintptr_t lo, uintptr_t hi)
{
    const uint64_t seq = g_seq.fetch_add(1, std::memory_order_relaxed) + 1;
    HistorySlot &h = g_hist[g_hist_head.fetch_add(1, std::memory_order_relaxed) % kHistory];
    h.seq.store(0, std::memory_order_release);
    h.lo.store(lo, std::memory_order_relaxed);
    h.hi.store(hi, std::memory_order_relaxed);
    h.kind.store(kind, std::memory_order_relaxed);
    h.op.store(op, std::memory_order_relaxed);
    h.seq.store(seq, std::memory_order_release);
    return seq;
}

} // namespace

bool range_note(RangeKind kind, const void *p, size_t bytes)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
    const uint64_t seq = history_line(kind, 1, lo, hi);
    for (LiveSlot &s : g_live) {
        uint32_t expect = 0;
        if (!s.state.compare_exchange_strong(expect, 1, std::memory_order_acquire)) continue;
        s.kind.store(kind, std::memory_order_relaxed);
        s.seq.store(seq, std::memory_order_relaxed);
        s.lo.store(lo, std::memory_order_relaxed);
        s.hi.store(hi, std::memory_order_relaxed);
        s.state.store(2, std::memory_order_release);
        return true;
    }
    g_overflow.store(true, std::memory_order_relaxed);
    return false;
}

bool range_forget(RangeKind kind, const void *p)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
    for (LiveSlot &s : g_live) {
        if (s.state.load(std::memory_order_acquire) != 2) continue;
        if (s.kind.load(std::memory_order_relaxed) != kind || s.lo.load(std::memory_order_relaxed) != lo) continue;
        uint32_t expect = 2;
        if (!s.state.compare_exchange_strong(expect, 1, std::memory_order_acquire)) continue; // somebody else forgot it first
        history_line(kind, 0, lo, s.hi.load(std::memory_order_relaxed));
        s.state.store(0, std::memory_order_release);
        return true;
    }
    return false;
}

void range_event(RangeKind kind, const void *p, size_t bytes)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
    history_line(kind, 1, lo, lo + bytes);
}

bool range_is_live(RangeKind kind, const void *p)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
    for (LiveSlot &s : g_live)
        if (s.state.load(std::memory_order_acquire) == 2 && s.kind.load(std::memory_order_relaxed) == kind &&
            s.lo.load(std::memory_order_relaxed) == lo)
            return true;
    return false;
}

size_t range_live_snapshot(RangeRecord *out, size_t cap)
{
    size_t n = 0;
    for (LiveSlot &s : g_live) {
        if (n == cap) break;
        if (s.state.load(std::memory_order_acquire) != 2) continue;
        out[n++] = RangeRecord{s.seq.load(std::memory_order_relaxed), s.lo.load(std::memory_order_relaxed),
                               s.hi.load(std::memory_order_relaxed), s.kind.load(std::memory_order_relaxed), 1u};
    }
    return n;
}

size_t range_history_snapshot(RangeRecord *out, size_t cap)
{
    const uint64_t head = g_hist_head.load(std::memory_order_acquire);
    const uint64_t first = head > kHistory ? head - kHistory : 0;
    size_t n = 0;
    for (uint64_t i = first; i < head && n < cap; ++i) {
        HistorySlot &h = g_hist[i % kHistory];
        const uint64_t seq = h.seq.load(std::memory_order_acquire);
        if (seq == 0) continue;
        out[n++] = RangeRecord{seq, h.lo.load(std::memory_order_relaxed), h.hi.load(std::memory_order_relaxed),
                               h.kind.load(std::memory_order_relaxed), h.op.load(std::memory_order_relaxed)};
    }
    return n;
}

bool range_overflowed() { return g_overflow.load(std::memory_order_relaxed); }

} // namespace nus

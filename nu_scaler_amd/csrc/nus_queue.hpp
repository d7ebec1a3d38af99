// nus_queue.hpp -- bounded frame queue feeding the path ("next" row).
// Mirrors the legacy FrameBuffer (Nu_scale/src/capture/frame_buffer.rs:11-50 and :52-100): a
// mutex-protected deque of shared frames with a fixed capacity; adding to a full queue drops the
// OLDEST frame; consumers take the latest frame (optionally waiting up to a timeout).
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

namespace nus {

struct QueuedFrame {
    std::vector<uint8_t> data; // tightly packed RGBA8
    uint32_t width = 0, height = 0;
    uint64_t sequence = 0; // 0-based index of the frame among all frames ever added
};

class FrameQueue {
public:
    explicit FrameQueue(size_t capacity) : capacity_(capacity ? capacity : 1) {}

    // add_frame (frame_buffer.rs:37-50): when full, pop the oldest first.  Returns the number of
    // frames dropped so far.
    uint64_t add(const uint8_t *rgba, uint32_t w, uint32_t h)
    {
        auto f = std::make_shared<QueuedFrame>();
        f->data.assign(rgba, rgba + (size_t)w * h * 4);
        f->width = w;
        f->height = h;
        std::lock_guard<std::mutex> lk(mu_);
        f->sequence = next_seq_++;
        if (frames_.size() >= capacity_) {
            frames_.pop_front();
            ++dropped_;
        }
        frames_.push_back(std::move(f));
        cv_.notify_all();
        return dropped_;
    }

    // get_latest_frame (:53-55) / get_latest_frame_timeout (:57-..): newest frame, or null.
    std::shared_ptr<QueuedFrame> latest(int64_t timeout_ms = 0)
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (frames_.empty() && timeout_ms > 0)
            cv_.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return !frames_.empty(); });
        return frames_.empty() ? nullptr : frames_.back();
    }

    // Oldest frame, removed from the queue (FIFO consumption for a stream that must not skip) -- but only if the
    // caller can take it: the size is checked against `max_bytes` UNDER the queue's lock, so a frame the caller's
    // buffer cannot hold stays queued (`too_big` is set) instead of being popped and lost.
    std::shared_ptr<QueuedFrame> pop(int64_t timeout_ms = 0, size_t max_bytes = SIZE_MAX, bool *too_big = nullptr)
    {
        if (too_big) *too_big = false;
        std::unique_lock<std::mutex> lk(mu_);
        if (frames_.empty() && timeout_ms > 0)
            cv_.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return !frames_.empty(); });
        if (frames_.empty()) return nullptr;
        if (frames_.front()->data.size() > max_bytes) {
            if (too_big) *too_big = true;
            return nullptr;
        }
        auto f = frames_.front();
        frames_.pop_front();
        return f;
    }

    size_t size() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return frames_.size();
    }
    size_t capacity() const { return capacity_; }
    uint64_t dropped() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return dropped_;
    }
    void clear()
    {
        std::lock_guard<std::mutex> lk(mu_);
        frames_.clear();
    }

private:
    const size_t capacity_;
    mutable std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<QueuedFrame>> frames_;
    uint64_t next_seq_ = 0, dropped_ = 0;
};

} // namespace nus

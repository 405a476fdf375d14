// Worker threads for the HOST side of large host-buffer calls: the copies between the caller's pageable arrays and the pinned
// staging blocks (rvll_api.hip, stream_host_batch and download_rows).  Plain C++17, no HIP: tests/test_copypool_native.py runs
// it on the CPU under ThreadSanitizer and AddressSanitizer.
//
// One pool per handle, made by the first call that needs it.  The workers sleep on a condition variable between calls; while
// a call is running (busy(true) .. busy(false)) they poll a counter instead — a wake-up through the futex costs tens of
// microseconds, a chunk's copy a hundred.  A ticket counts the pieces of a copy that are still to do; the caller owns it and
// must not let it go out of scope before wait() has returned.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace rvll {

class CopyPool {
public:
    struct Ticket { std::atomic<int> left{0}; };
    explicit CopyPool(int n)
    {
        try {
            for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
        } catch (...) {                                 // a thread could not be started: stop the ones that were
            shutdown();
            throw;
        }
    }
    ~CopyPool() { shutdown(); }
    void shutdown()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) if (t.joinable()) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // while a call is running the workers poll for work; between calls they sleep
    void busy(bool on)
    {
        { std::lock_guard<std::mutex> g(m_); busy_.store(on, std::memory_order_relaxed); }
        if (on) cv_.notify_all();
    }
    // dst <- src in page-aligned pieces, one per worker at most and none below 128 KB; the ticket counts the pieces still to do
    void copy(void* dst, const void* src, size_t bytes, Ticket* t)
    {
        if (!bytes) return;
        const size_t pieces = std::max<size_t>(1, std::min<size_t>(workers_.size(), bytes / (128u << 10)));
        const size_t step = ((bytes + pieces - 1) / pieces + 4095) & ~(size_t)4095;
        const int n = (int)((bytes + step - 1) / step);
        t->left.fetch_add(n, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t off = 0; off < bytes; off += step)
                q_.push_back({static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, std::min(step, bytes - off), t});
            queued_.store((int)q_.size(), std::memory_order_release);
        }
        if (!busy_.load(std::memory_order_relaxed)) cv_.notify_all();
    }
    static void wait(Ticket* t)
    {
        for (unsigned spins = 0; t->left.load(std::memory_order_acquire) > 0; ++spins)
            if (spins < 4096) __builtin_ia32_pause(); else std::this_thread::yield();
    }
private:
    struct Task { char* dst; const char* src; size_t bytes; Ticket* t; };
    void run()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!q_.empty()) {
                const Task k = q_.front();
                q_.pop_front();
                queued_.store((int)q_.size(), std::memory_order_release);
                lk.unlock();
                memcpy(k.dst, k.src, k.bytes);
                k.t->left.fetch_sub(1, std::memory_order_release);
                lk.lock();
            } else if (stop_) {
                return;
            } else if (busy_.load(std::memory_order_relaxed)) {
                lk.unlock();                    // (poll without the lock: eight idle workers taking it every microsecond slowed the caller)
                for (int i = 0; i < 4096 && queued_.load(std::memory_order_acquire) == 0; ++i) __builtin_ia32_pause();
                lk.lock();
            } else {
                cv_.wait(lk);
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    std::vector<std::thread> workers_;
    std::atomic<int> queued_{0};
    bool stop_ = false;                 // (under m_)
    std::atomic<bool> busy_{false};
};

}  // namespace rvll

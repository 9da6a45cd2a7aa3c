// shard_workers.hpp -- threads that outlive the calls (the single-process multi-GPU handle, multigpu.cpp).
// Pure host code, no HIP: what a worker does when it starts (make its shard's device current) is a hook, so that
// tests/c/shard_workers_selftest.cpp can run the pool under ThreadSanitizer on a machine without a GPU.
#ifndef RSP_SHARD_WORKERS_HPP
#define RSP_SHARD_WORKERS_HPP

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace rsp {

inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

inline void futex_wait(std::atomic<uint32_t>* word, uint32_t expected) {
    (void)syscall(SYS_futex, (uint32_t*)word, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* word) {
    (void)syscall(SYS_futex, (uint32_t*)word, FUTEX_WAKE_PRIVATE, INT32_MAX, nullptr, nullptr, 0);
}
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");

inline int spin_us_setting() {
    static const int v = [] {
        const char* s = getenv("RSP_MCSC_SPIN_US");
        const int n = (s && s[0]) ? atoi(s) : 50;
        return n < 0 ? 0 : n;
    }();
    return v;
}

// Threads that outlive the calls: worker j (1-based shard index) runs fn(ctx, j) whenever the owner bumps the
// generation; the owner runs fn(ctx, 0) itself and then waits for `pending` to drain.
class ShardWorkers {
public:
    using Fn = void (*)(void* ctx, int shard);
    ShardWorkers() = default;
    ShardWorkers(const ShardWorkers&) = delete;
    ShardWorkers& operator=(const ShardWorkers&) = delete;
    ~ShardWorkers() { stop(); }

    // one thread per entry of `devices` (the device of shards 1 .. n); false: no threads to be had (nothing left running)
    bool start(const std::vector<int>& devices, void (*on_start)(int device) = nullptr) noexcept {
        try {
            on_start_ = on_start;
            devices_ = devices;
            threads_.reserve(devices.size());
            for (size_t j = 0; j < devices.size(); ++j) threads_.emplace_back([this, j] { loop((int)j + 1); });
            return true;
        } catch (...) {
            stop();
            return false;
        }
    }
    int size() const { return (int)threads_.size(); }

    void run(Fn fn, void* ctx) noexcept {
        fn_ = fn;
        ctx_ = ctx;
        pending_.store((uint32_t)threads_.size(), std::memory_order_relaxed);
        generation_.fetch_add(1, std::memory_order_seq_cst);
        if (parked_.load(std::memory_order_seq_cst) > 0) futex_wake_all(&generation_);
        fn(ctx, 0);
        unsigned spins = 0;
        while (pending_.load(std::memory_order_acquire) != 0) {
            if ((++spins & 0xfff) == 0) std::this_thread::yield();   // (fewer cores than shards: let a worker run)
            else cpu_relax();
        }
    }

private:
    void stop() noexcept {
        if (threads_.empty()) return;
        quit_.store(true, std::memory_order_release);
        generation_.fetch_add(1, std::memory_order_seq_cst);
        futex_wake_all(&generation_);
        for (auto& t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
    }

    void loop(int shard) noexcept {
        if (on_start_) on_start_(devices_[(size_t)shard - 1]);   // (the handle: hipSetDevice -- a thread's current device stays: set once, not per call)
        uint32_t seen = 0;
        const double spin_us = (double)spin_us_setting();
        for (;;) {
            uint32_t g;
            double t0 = now_us();
            unsigned spins = 0;
            while ((g = generation_.load(std::memory_order_acquire)) == seen) {
                if ((++spins & 63) == 0 && now_us() - t0 >= spin_us) {
                    parked_.fetch_add(1, std::memory_order_seq_cst);
                    futex_wait(&generation_, seen);   // (returns at once if the generation has moved meanwhile)
                    parked_.fetch_sub(1, std::memory_order_seq_cst);
                    t0 = now_us();
                } else {
                    cpu_relax();
                }
            }
            seen = g;
            if (quit_.load(std::memory_order_acquire)) return;
            fn_(ctx_, shard);
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

    std::vector<std::thread> threads_;
    std::vector<int> devices_;
    void (*on_start_)(int) = nullptr;
    Fn fn_ = nullptr;
    void* ctx_ = nullptr;
    alignas(64) std::atomic<uint32_t> generation_{0};
    alignas(64) std::atomic<uint32_t> pending_{0};
    alignas(64) std::atomic<int> parked_{0};
    std::atomic<bool> quit_{false};
};

}  // namespace rsp
#endif

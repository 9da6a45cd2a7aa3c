// CPU self-test of the worker pool behind rsp_mcsc_column_sums (csrc/shard_workers.hpp): the threads that stay parked
// between the calls of the single-process multi-GPU handle.  No HIP: the pool is pure host code.  Built by
// tests/test_capi_nogpu.py plain and under ThreadSanitizer.  What it holds the pool to:
//   * every run() executes fn(ctx, k) exactly once for k = 0 .. n, and returns only after all of them have;
//   * what the workers wrote is visible to the caller after run() (the slices they copy into the caller's vector);
//   * workers that have PARKED on the futex (RSP_MCSC_SPIN_US=0: at once) are woken by the next run -- no lost wake-up
//     whatever the timing (runs back to back, runs after pauses, runs racing a worker on its way to sleep);
//   * the on-start hook runs once per worker with its own device ordinal, before its first piece of work;
//   * destruction joins everything, also right after a run and also without any run; pools come and go.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../../rcppsparse_amd/csrc/shard_workers.hpp"

#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                      \
        }                                                                      \
    } while (0)

namespace {
std::atomic<int> g_started{0};
std::atomic<long> g_device_sum{0};
void on_start(int device) {
    g_started.fetch_add(1);
    g_device_sum.fetch_add(device);
}

struct Ctx {
    std::vector<long> slot;       // plain memory: one element per shard, written by that shard's thread only
    std::vector<int> calls;       // how often fn ran for shard k in the current run
    long round = 0;
    int busy_us = 0;
};
void work(void* vctx, int k) {
    Ctx* c = (Ctx*)vctx;
    if (c->busy_us > 0 && (k % 3) == 1) std::this_thread::sleep_for(std::chrono::microseconds(c->busy_us));
    c->slot[(size_t)k] = c->round * 1000 + k;
    c->calls[(size_t)k] += 1;
}
}  // namespace

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
    std::mt19937 rng(12345);
    for (int nworkers : {1, 2, 7, 15}) {
        g_started = 0;
        g_device_sum = 0;
        std::vector<int> devices;
        long want_sum = 0;
        for (int j = 0; j < nworkers; ++j) {
            devices.push_back(100 + j);
            want_sum += 100 + j;
        }
        {
            rsp::ShardWorkers pool;
            CHECK(pool.start(devices, on_start));
            CHECK(pool.size() == nworkers);
            Ctx c;
            c.slot.assign((size_t)nworkers + 1, -1);
            c.calls.assign((size_t)nworkers + 1, 0);
            for (int r = 0; r < rounds; ++r) {
                c.round = r;
                c.busy_us = (r % 97 == 0) ? 30 : 0;
                for (auto& n : c.calls) n = 0;
                // pauses of every length around the workers' way to sleep: none, shorter than a spin window, longer
                const unsigned pick = rng() % 16;
                if (pick == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
                else if (pick == 1) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 60));
                else if (pick == 2) std::this_thread::yield();
                pool.run(work, &c);
                for (int k = 0; k <= nworkers; ++k) {
                    CHECK(c.calls[(size_t)k] == 1);
                    CHECK(c.slot[(size_t)k] == (long)r * 1000 + k);
                }
            }
            CHECK(g_started.load() == nworkers);        // once per worker, and before its first piece of work
            CHECK(g_device_sum.load() == want_sum);
        }   // (joined here)
    }
    // pools that are made and dropped without a run, and right after one
    for (int rep = 0; rep < 200; ++rep) {
        rsp::ShardWorkers pool;
        CHECK(pool.start({0, 1, 2}));
        if (rep & 1) {
            Ctx c;
            c.slot.assign(4, -1);
            c.calls.assign(4, 0);
            pool.run(work, &c);
            CHECK(c.calls[3] == 1);
        }
    }
    std::printf("shard workers selftest ok\n");
    return 0;
}

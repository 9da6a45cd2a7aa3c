// multigpu.cpp -- column-range partitioner (pure integer, host) and the RCCL
// gatherv of per-shard sums.  Nothing like this exists in the reference (it has
// no distributed code at all); columns are independent
// units of reference src/example.cpp:28, so contiguous column ranges shard the
// path with exactly one exchange step: the gather of disjoint output slices.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rcppsparse_hip.h"
#include "colsums_kernels.h"
#include "shard_workers.hpp"

static_assert(RSP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace rsp {
int fail(int code, const char* fmt, ...);   // capi.hip: sets rsp_last_error() text
}
using rsp::fail;

namespace {

// Runs work(0..G-1) with one host thread per shard (shard 0 on the calling thread, whose current
// HIP device is put back afterwards).  Started threads are always joined, also when creating a
// later one throws; nothing escapes: std::bad_alloc / std::system_error become `false`.
template <class Work>
bool run_shards(int G, Work&& work) noexcept {
    struct Joiner {
        std::vector<std::thread> threads;
        ~Joiner() {
            for (auto& t : threads)
                if (t.joinable()) t.join();
        }
    };
    bool ok = true;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    try {
        Joiner j;
        j.threads.reserve(G > 1 ? (size_t)G - 1 : 0);
        for (int k = 1; k < G; ++k) j.threads.emplace_back([&work, k] { work(k); });
        work(0);
    } catch (...) {   // (the Joiner has joined whatever was started)
        ok = false;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok;
}

}  // namespace

struct rsp_comm {
    ncclComm_t comm;
    int nranks, rank, device;
};

extern "C" {

int rsp_partition_columns(const int32_t* p, int32_t ncol, int32_t nparts, int32_t* bounds) {
    if (!p || !bounds || ncol < 0 || nparts <= 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_partition_columns");
    const int64_t nnz = p[ncol];
    bounds[0] = 0;
    for (int32_t k = 1; k < nparts; ++k) {
        // first column c with p[c] >= k*nnz/nparts  (lower_bound on p[0..ncol])
        const int64_t target = (int64_t)(((__int128)k * nnz) / nparts);
        int32_t lo = 0, hi = ncol;   // answer in [lo, hi]; p[ncol] = nnz >= target
        while (lo < hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (p[mid] >= target) hi = mid; else lo = mid + 1;
        }
        bounds[k] = lo < bounds[k - 1] ? bounds[k - 1] : lo;
    }
    bounds[nparts] = ncol;
    return RSP_OK;
}

int rsp_rebase_offsets(const int32_t* p, int32_t c0, int32_t c1, int32_t* p_local) {
    if (!p || !p_local || c0 < 0 || c1 < c0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_rebase_offsets");
    const int32_t base = p[c0];
    for (int32_t j = 0; j <= c1 - c0; ++j) p_local[j] = p[c0 + j] - base;
    return RSP_OK;
}

// ---- one-shot host path over several GPUs -----------------------
// The one-shot call is PCIe-bound, so with G GPUs the columns are cut
// into nnz-balanced ranges and every range travels over its own GPU's host link: one host
// thread per shard uploads x[p[c0]:p[c1]] and the rebased offsets, runs the same kernels,
// and copies its slice of the sums straight into sums + c0.  No collective is needed: the
// result lives in host memory.  `devices` may repeat an ordinal (several shards on one GPU).
int rsp_column_sums_host_multi(const double* x, const int32_t* p, int32_t ncol, int64_t nnz, double* sums,
                               const int* devices, int ndevices) try {
    if (!p || (nnz > 0 && !x) || (ncol > 0 && !sums) || ncol < 0 || nnz < 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_column_sums_host_multi");
    int visible = 0;
    if (rsp::process_was_forked_after_gpu_use())
        return fail(RSP_ERR_NO_DEVICE, "this process was forked from one that had already used the GPU: the HIP runtime does not survive a fork");
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return fail(RSP_ERR_NO_DEVICE, "no HIP device available");
    }
    std::vector<int> devs;
    if (devices && ndevices > 0) {
        devs.assign(devices, devices + ndevices);
    } else {
        for (int d = 0; d < visible; ++d) devs.push_back(d);
    }
    for (int d : devs)
        if (d < 0 || d >= visible) return fail(RSP_ERR_BAD_ARG, "device %d out of range [0, %d)", d, visible);
    if (ncol == 0) return RSP_OK;
    // same contract on p as the single-device entry point
    if (p[0] != 0 || p[ncol] != nnz) return fail(RSP_ERR_BAD_ARG, "p[0] must be 0 and p[ncol] must equal nnz");
    int bad = 0;
    for (int32_t c = 0; c < ncol; ++c) bad |= (p[c + 1] < p[c]);
    if (bad) return fail(RSP_ERR_BAD_ARG, "p[] is not non-decreasing");

    const int G = (int)devs.size();
    std::vector<int32_t> bounds((size_t)G + 1);
    if (int rc = rsp_partition_columns(p, ncol, G, bounds.data())) return rc;

    std::vector<int> status((size_t)G, RSP_OK);
    std::vector<std::string> message((size_t)G);
    auto work = [&](int k) noexcept {
        const int32_t c0 = bounds[k], c1 = bounds[k + 1];
        const int32_t nc = c1 - c0;
        if (nc == 0) return;
        const int64_t n0 = p[c0], nk = (int64_t)p[c1] - p[c0];
        double *d_x = nullptr, *d_out = nullptr;   // released on every path below
        int32_t* d_p = nullptr;
        void* d_ws = nullptr;
        hipStream_t st = nullptr;
        try {
            std::vector<int32_t> pk((size_t)nc + 1);
            for (int32_t j = 0; j <= nc; ++j) pk[j] = p[c0 + j] - (int32_t)n0;
            auto check = [&](hipError_t e, const char* what) {
                if (e != hipSuccess && status[k] == RSP_OK) {
                    status[k] = RSP_ERR_HIP;
                    message[k] = std::string(what) + ": " + hipGetErrorString(e);
                }
                return e == hipSuccess;
            };
            const size_t xbytes = (((size_t)nk * 8 + 15) & ~(size_t)15) + 16;
            bool ok = check(hipSetDevice(devs[k]), "hipSetDevice");
            const size_t wsb = rsp_column_sums_workspace_bytes(nc, nk);
            ok = ok && check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
            ok = ok && check(hipMalloc((void**)&d_x, xbytes), "hipMalloc x");
            ok = ok && check(hipMalloc((void**)&d_p, ((size_t)nc + 1) * 4), "hipMalloc p");
            ok = ok && check(hipMalloc((void**)&d_out, (size_t)nc * 8), "hipMalloc out");
            ok = ok && check(hipMalloc(&d_ws, wsb), "hipMalloc workspace");
            if (ok && nk > 0)
                ok = check(hipMemcpyAsync(d_x, x + n0, (size_t)nk * 8, hipMemcpyHostToDevice, st), "H2D x");
            ok = ok && check(hipMemcpyAsync(d_p, pk.data(), ((size_t)nc + 1) * 4, hipMemcpyHostToDevice, st), "H2D p");
            if (ok) {
                const int rc = rsp::column_sums_general(d_x, d_p, nc, nk, d_out, d_ws, wsb, st);   // (a one-shot call: nothing to plan for)
                if (rc != RSP_OK) {
                    status[k] = rc;
                    message[k] = rsp_last_error();   // this thread's message
                    ok = false;
                }
            }
            ok = ok && check(hipMemcpyAsync(sums + c0, d_out, (size_t)nc * 8, hipMemcpyDeviceToHost, st), "D2H sums");
            if (st) check(hipStreamSynchronize(st), "hipStreamSynchronize");   // (pk is still alive here)
        } catch (...) {   // a host allocation of this shard failed (its offsets or an error text)
            if (st) (void)hipStreamSynchronize(st);
            status[k] = RSP_ERR_ALLOC;
        }
        if (d_x) (void)hipFree(d_x);
        if (d_p) (void)hipFree(d_p);
        if (d_out) (void)hipFree(d_out);
        if (d_ws) (void)hipFree(d_ws);
        if (st) (void)hipStreamDestroy(st);
    };
    if (!run_shards(G, work)) return fail(RSP_ERR_ALLOC, "out of host memory or threads while running the shards");
    for (int k = 0; k < G; ++k)
        if (status[k] != RSP_OK) return fail(status[k], "shard %d on device %d: %s", k, devs[k], message[k].c_str());
    return RSP_OK;
} catch (...) {   // std::bad_alloc from the bookkeeping vectors: nothing was started yet
    return fail(RSP_ERR_ALLOC, "out of host memory in rsp_column_sums_host_multi");
}

}  // extern "C"

// ---- resident matrix spread over several GPUs: upload once, sum many (f2) --------------------------
// This is the multi-GPU path an R session reaches: ONE process (reference src/RcppExports.cpp:16-24 runs on the R main
// thread), G devices.  A shard of C4 (1.25e8 entries) sums in ~150 us, and ">= 6x at 8 GPUs" leaves ~50 us for
// everything else, so nothing per call may cost what a thread creation, a stream creation or an allocation costs.
// The handle therefore keeps, per shard: the resident rsp_csc (its own stream, output and plan), and per handle: ONE
// page-locked host vector of ncol doubles that every shard's slice lands in, optionally G - 1 worker threads that stay
// parked between calls (each with its shard's device current for good), and -- for the RCCL gather -- the communicators
// of one ncclCommInitAll.  A call is, per shard: enqueue the column-sum launches, enqueue the slice's way home and an
// event behind it, poll that event (hipStreamSynchronize costs ~9 us even on a drained stream), copy the slice into the
// caller's vector.
//   launch  serial  : the calling thread issues every shard's kernels, then every shard's copy command + event, and
//                     then polls the events in turn (the shards are handed to the caller in the order they finish);
//           workers : shard 0 on the calling thread, shard k on its parked thread -- enqueues, waits and the copies
//                     into the caller's (pageable) vector all run side by side.  A worker spins for a short while
//                     after a call (RSP_MCSC_SPIN_US, default 50) so that calls in a loop find it awake, then sleeps
//                     on a futex: an idle R session burns nothing.
//   gather  D2H     : hipMemcpyAsync of every slice over its own device's host link into the page-locked vector;
//           BLIT    : the same trip by a copy kernel of the library behind the shard's kernels (default with a device per shard);
//           RCCL    : grouped ncclSend / ncclRecv of the slices to shard 0's device over xGMI (the collective
//                     BASELINE.json's north_star names), then one D2H of the whole vector;
//           STORES  : the kernels take the page-locked vector (+ the shard's first column) as their output: no copy
//                     command at all, the result stores travel over the host link as they are made.
// All of them give the bits of the per-shard device calls: the same launches on the same data.
namespace {

using rsp::ShardWorkers;
using rsp::now_us;
using rsp::cpu_relax;

// a worker's one-time set-up: its shard's device current for good, and its capture mode relaxed -- the event queries of a
// call are among the runtime calls that would invalidate a stream capture another thread of the process runs in global mode
void make_device_current(int device) {
    (void)hipSetDevice(device);
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError();
}

struct RelaxedCapture {   // the same for the calling thread, for the duration of a call
    hipStreamCaptureMode prev = hipStreamCaptureModeRelaxed;
    bool ok;
    RelaxedCapture() : ok(hipThreadExchangeStreamCaptureMode(&prev) == hipSuccess) {
        if (!ok) (void)hipGetLastError();
    }
    ~RelaxedCapture() {
        if (ok && hipThreadExchangeStreamCaptureMode(&prev) != hipSuccess) (void)hipGetLastError();
    }
};

struct ShardState {
    rsp::CscView view;        // device, stream, own output of the resident shard
    hipEvent_t done = nullptr;   // recorded behind the shard's last command of a call (no timing: as light as an event gets)
    int status = RSP_OK;
    char message[256] = "";
    double t_begin = 0, t_enqueued = 0, t_done = 0, t_copied = 0;   // host clock of the last call, microseconds
};

}  // namespace

struct rsp_mcsc {
    std::vector<rsp_csc_t> shards;
    std::vector<int32_t> bounds;   // column range of shard k: [bounds[k], bounds[k+1])
    std::vector<int> devices;
    int32_t nrow = 0, ncol = 0;
    bool has_rows = false;         // uploaded with i[]: the row-wise entries are available
    // what a column-sum call needs, made once
    std::vector<ShardState> st;
    double* h_result = nullptr;    // page-locked, ncol doubles: every shard's slice lands here
    double* d_result_view = nullptr;   // the same memory as the devices address it (RSP_GATHER_BLIT / _STORES)
    int gather = RSP_GATHER_D2H;
    int launch = RSP_LAUNCH_SERIAL;
    ShardWorkers* workers = nullptr;   // made on the first call that wants them
    bool workers_failed = false;
    std::vector<ncclComm_t> comms;     // RSP_GATHER_RCCL: one ncclCommInitAll over the shards' devices
    double* d_gathered = nullptr;      // ... and the gathered vector on shard 0's device
    double t_call_begin = 0, t_call_end = 0;
    // row sums reduced ON THE DEVICES (round 6): per shard its partial vector, the slices it receives, its reduced slice
    struct RowShard {
        double* d_partial = nullptr;    // nrow doubles: the shard's own columns' row sums
        double* d_incoming = nullptr;   // G x len: slice k of this shard's rows from every shard's partial vector
        double* d_reduced = nullptr;    // len: this shard's slice of the result
        hipEvent_t computed = nullptr, done = nullptr;
    };
    std::vector<RowShard> rows;
    double* h_rows = nullptr;        // page-locked, nrow doubles: the add kernels write their reduced slices straight into it
    double* d_rows_view = nullptr;   // ... as the devices address it (nullptr: the slices come home by a copy instead)
    bool rows_ready = false, rows_failed = false;
    // R code forks (parallel::mclapply): a child inherits this pointer but neither the worker threads nor a usable GPU context.
    // A call there must fail with a message, not wait for threads that do not exist.
    pid_t owner = getpid();
};

namespace {

void shard_fail(ShardState& s, int code, const char* what, const char* detail) noexcept {
    if (s.status != RSP_OK) return;
    s.status = code;
    snprintf(s.message, sizeof(s.message), "%s: %s", what, detail ? detail : "");
}

struct ColumnCall {
    rsp_mcsc* h;
    double* sums;
    bool means;
};

// shard k, step 1: the column-sum launches on the shard's own stream (its device current)
void shard_launch(ColumnCall* c, int k) noexcept {
    rsp_mcsc* h = c->h;
    ShardState& s = h->st[(size_t)k];
    s.status = RSP_OK;
    s.t_begin = now_us();
    const int32_t c0 = h->bounds[(size_t)k], nc = h->bounds[(size_t)k + 1] - c0;
    if (nc > 0) {
        double* d_out = nullptr;   // the shard's own output
        if (h->gather == RSP_GATHER_STORES) d_out = h->d_result_view + c0;
        else if (h->gather == RSP_GATHER_RCCL && k == 0) d_out = h->d_gathered + c0;   // shard 0's slice is in place
        const int rc = rsp::csc_enqueue_columns(h->shards[(size_t)k], c->means, d_out);
        if (rc != RSP_OK) shard_fail(s, rc, "column sums", rsp_last_error());
    }
    s.t_enqueued = now_us();
}

// step 2: the slice's way to the page-locked vector, behind the launches on the same stream
void shard_send(ColumnCall* c, int k) noexcept {
    rsp_mcsc* h = c->h;
    ShardState& s = h->st[(size_t)k];
    const int32_t c0 = h->bounds[(size_t)k], nc = h->bounds[(size_t)k + 1] - c0;
    if (nc > 0 && s.status == RSP_OK && h->gather == RSP_GATHER_D2H) {
        const hipError_t e = hipMemcpyAsync(h->h_result + c0, s.view.d_out, (size_t)nc * 8, hipMemcpyDeviceToHost, s.view.stream);
        if (e != hipSuccess) shard_fail(s, RSP_ERR_HIP, "D2H of the slice", hipGetErrorString(e));
    }
    if (nc > 0 && s.status == RSP_OK && h->gather == RSP_GATHER_BLIT) {
        const hipError_t e = rsp::launch_copy_f64(s.view.d_out, h->d_result_view + c0, nc, s.view.stream);
        if (e != hipSuccess) shard_fail(s, RSP_ERR_HIP, "copy kernel of the slice", hipGetErrorString(e));
    }
    if (nc > 0) {
        const hipError_t e = hipEventRecord(s.done, s.view.stream);
        if (e != hipSuccess) shard_fail(s, RSP_ERR_HIP, "hipEventRecord", hipGetErrorString(e));
    }
}

// step 3: has the shard's event passed?  (A look costs a fraction of a microsecond; hipStreamSynchronize on a stream that
// has ALREADY drained costs ~9 us on this runtime -- eight of them in a row were most of a call, profiles/r06_mcsc_overhead.md.)
// true: the shard is done with (its slice handed to the caller, or its failure recorded).
bool shard_poll(ColumnCall* c, int k) noexcept {
    rsp_mcsc* h = c->h;
    ShardState& s = h->st[(size_t)k];
    const int32_t c0 = h->bounds[(size_t)k], nc = h->bounds[(size_t)k + 1] - c0;
    if (nc > 0) {
        const hipError_t q = s.status == RSP_OK ? hipEventQuery(s.done) : hipErrorUnknown;
        if (q == hipErrorNotReady) return false;
        if (q != hipSuccess) {
            (void)hipGetLastError();
            if (s.status == RSP_OK) shard_fail(s, RSP_ERR_HIP, "waiting for the shard", hipGetErrorString(q));
            (void)hipStreamSynchronize(s.view.stream);   // nothing of a failed call stays in flight
        }
    }
    s.t_done = now_us();
    if (nc > 0 && s.status == RSP_OK && c->sums != h->h_result && h->gather != RSP_GATHER_NONE)
        memcpy(c->sums + c0, h->h_result + c0, (size_t)nc * 8);
    s.t_copied = now_us();
    return true;
}

void shard_whole(void* ctx, int k) noexcept {
    shard_launch((ColumnCall*)ctx, k);
    shard_send((ColumnCall*)ctx, k);
    while (!shard_poll((ColumnCall*)ctx, k)) cpu_relax();
}
void shard_copy_only(void* ctx, int k) noexcept {   // RCCL gather: the whole vector is in h_result; the slices go out side by side
    ColumnCall* c = (ColumnCall*)ctx;
    rsp_mcsc* h = c->h;
    const int32_t c0 = h->bounds[(size_t)k], nc = h->bounds[(size_t)k + 1] - c0;
    if (nc > 0 && c->sums != h->h_result) memcpy(c->sums + c0, h->h_result + c0, (size_t)nc * 8);
    h->st[(size_t)k].t_copied = now_us();
}

bool ensure_workers(rsp_mcsc* h) noexcept {
    if (h->workers) return true;
    if (h->workers_failed || h->shards.size() < 2) return false;
    ShardWorkers* w = new (std::nothrow) ShardWorkers();
    std::vector<int> devs;
    try {
        devs.assign(h->devices.begin() + 1, h->devices.end());
    } catch (...) {
        delete w;
        w = nullptr;
    }
    if (!w || !w->start(devs, make_device_current)) {
        delete w;
        h->workers_failed = true;   // (no threads to be had: the calling thread does the work, now and later)
        return false;
    }
    h->workers = w;
    return true;
}

void mcsc_release_rccl(rsp_mcsc* h) noexcept {
    for (size_t k = 0; k < h->comms.size(); ++k)
        if (h->comms[k]) {
            (void)hipSetDevice(h->devices[k]);
            (void)ncclCommDestroy(h->comms[k]);
        }
    h->comms.clear();
    if (h->d_gathered) {
        (void)hipSetDevice(h->devices[0]);
        (void)hipFree(h->d_gathered);
        h->d_gathered = nullptr;
    }
}

int default_launch(int nshards) {
    const char* s = getenv("RSP_MCSC_LAUNCH");
    if (s && !strcmp(s, "serial")) return RSP_LAUNCH_SERIAL;
    if (s && !strcmp(s, "workers")) return RSP_LAUNCH_WORKERS;
    // measured (profiles/r06_mcsc_overhead.json): from three shards on, eight enqueues one after the other cost the
    // last shard more than a parked thread's wake-up does
    return nshards >= 3 ? RSP_LAUNCH_WORKERS : RSP_LAUNCH_SERIAL;
}

// the per-handle state of the column-sum calls, once the shards exist (h->shards, bounds, devices filled)
int mcsc_prepare_on_devices(rsp_mcsc* h);
int mcsc_prepare(rsp_mcsc* h) {   // (leaves the calling thread's current device as it found it)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    const int rc = mcsc_prepare_on_devices(h);
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
int mcsc_prepare_on_devices(rsp_mcsc* h) {
    const size_t G = h->shards.size();
    try {
        h->st.assign(G, ShardState());
    } catch (...) {
        return fail(RSP_ERR_ALLOC, "out of host memory");
    }
    for (size_t k = 0; k < G; ++k) {
        if (int rc = rsp::csc_view(h->shards[k], &h->st[k].view)) return rc;
        hipError_t e = hipSetDevice(h->devices[k]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->st[k].done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            h->st[k].done = nullptr;
            return fail(RSP_ERR_HIP, "event of shard %zu: %s", k, hipGetErrorString(e));
        }
    }
    const size_t bytes = h->ncol > 0 ? (size_t)h->ncol * 8 : 8;
    // portable: page-locked for EVERY device of the process (each shard's copy engine writes its own slice)
    hipError_t e = hipHostMalloc((void**)&h->h_result, bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        h->h_result = nullptr;
        return fail(RSP_ERR_ALLOC, "page-locked result vector (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, h->h_result, 0) == hipSuccess) h->d_result_view = (double*)dv;
    else (void)hipGetLastError();
    // default gather: with a DEVICE per shard a copy kernel behind the shard's kernels brings the slice home (1 MB: 19 us
    // against 36 us for the runtime's copy command, profiles/r06_mcsc_overhead.md section 4); shards that share a device
    // share its CUs and its one host link, where the copy command is no worse
    bool distinct = h->d_result_view != nullptr;
    for (size_t a = 0; a < G && distinct; ++a)
        for (size_t b = a + 1; b < G; ++b)
            if (h->devices[a] == h->devices[b]) distinct = false;
    if (const char* g = getenv("RSP_MCSC_GATHER")) {
        if (!strcmp(g, "d2h")) distinct = false;
        else if (!strcmp(g, "blit") && h->d_result_view) distinct = true;
    }
    h->gather = distinct ? RSP_GATHER_BLIT : RSP_GATHER_D2H;
    h->launch = default_launch((int)G);
    return RSP_OK;
}

}  // namespace

extern "C" {

int rsp_mcsc_free(rsp_mcsc_t h) {
    if (!h) return RSP_OK;
    if (h->owner != getpid()) return RSP_OK;   // (a forked child: nothing here is its to release; the parent's copy lives on)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    delete h->workers;   // (joins the parked threads)
    h->workers = nullptr;
    for (rsp_csc_t s : h->shards) rsp_csc_free(s);   // waits for each shard's stream
    for (ShardState& s : h->st)
        if (s.done) (void)hipEventDestroy(s.done);
    if (h->h_rows) (void)hipHostFree(h->h_rows);
    for (size_t k = 0; k < h->rows.size(); ++k) {
        rsp_mcsc::RowShard& r = h->rows[k];
        (void)hipSetDevice(h->devices[k]);
        if (r.d_partial) (void)hipFree(r.d_partial);
        if (r.d_incoming) (void)hipFree(r.d_incoming);
        if (r.d_reduced) (void)hipFree(r.d_reduced);
        if (r.computed) (void)hipEventDestroy(r.computed);
        if (r.done) (void)hipEventDestroy(r.done);
    }
    mcsc_release_rccl(h);
    if (h->h_result) (void)hipHostFree(h->h_result);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete h;
    return RSP_OK;
}

static int mcsc_upload(const double* x, const int32_t* i, const int32_t* p, int32_t nrow, int32_t ncol, int64_t nnz,
                       const int* devices, int ndevices, rsp_mcsc_t* handle) {
    if (!handle) return fail(RSP_ERR_BAD_ARG, "handle is null");
    *handle = nullptr;
    if (!p || (nnz > 0 && !x) || ncol < 0 || nnz < 0) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_mcsc_upload");
    int visible = 0;
    if (rsp::process_was_forked_after_gpu_use())
        return fail(RSP_ERR_NO_DEVICE, "this process was forked from one that had already used the GPU: the HIP runtime does not survive a fork");
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return fail(RSP_ERR_NO_DEVICE, "no HIP device available");
    }
    rsp_mcsc* h = nullptr;
    try {
        std::vector<int> devs;
        if (devices && ndevices > 0) devs.assign(devices, devices + ndevices);
        else for (int d = 0; d < visible; ++d) devs.push_back(d);
        for (int d : devs)
            if (d < 0 || d >= visible) return fail(RSP_ERR_BAD_ARG, "device %d out of range [0, %d)", d, visible);
        if (ncol > 0 && (p[0] != 0 || p[ncol] != nnz))
            return fail(RSP_ERR_BAD_ARG, "p[0] must be 0 and p[ncol] must equal nnz");
        const int G = (int)devs.size();
        h = new rsp_mcsc();
        h->nrow = nrow;
        h->ncol = ncol;
        h->has_rows = i != nullptr || nnz == 0;
        h->devices = devs;
        h->bounds.assign((size_t)G + 1, 0);
        h->shards.assign((size_t)G, nullptr);
        if (int rc = rsp_partition_columns(p, ncol, G, h->bounds.data())) {
            delete h;
            return rc;
        }
        // one host thread per shard: every shard goes over its own GPU's host link (an upload is milliseconds to
        // seconds of copying: the threads of this ONE call are noise beside it)
        std::vector<int> status((size_t)G, RSP_OK);
        std::vector<std::string> message((size_t)G);
        auto work = [&](int k) noexcept {
            try {
                const int32_t c0 = h->bounds[k], c1 = h->bounds[k + 1];
                std::vector<int32_t> pk((size_t)(c1 - c0) + 1);
                for (int32_t j = 0; j <= c1 - c0; ++j) pk[j] = p[c0 + j] - p[c0];
                const int64_t nk = (int64_t)p[c1] - p[c0];
                status[k] = rsp_csc_upload(nk ? x + p[c0] : x, (i && nk) ? i + p[c0] : nullptr, pk.data(), nrow,
                                           c1 - c0, nk, devs[k], &h->shards[k]);
                if (status[k] != RSP_OK) message[k] = rsp_last_error();
            } catch (...) {   // (a shard that was uploaded stays in h->shards and is freed below)
                status[k] = RSP_ERR_ALLOC;
            }
        };
        if (!run_shards(G, work)) {
            rsp_mcsc_free(h);
            return fail(RSP_ERR_ALLOC, "out of host memory or threads while uploading the shards");
        }
        for (int k = 0; k < G; ++k)
            if (status[k] != RSP_OK) {
                const int rc = fail(status[k], "shard %d on device %d: %s", k, devs[k], message[k].c_str());
                rsp_mcsc_free(h);
                return rc;
            }
        if (int rc = mcsc_prepare(h)) {
            rsp_mcsc_free(h);
            return rc;
        }
    } catch (...) {
        if (h) rsp_mcsc_free(h);
        return fail(RSP_ERR_ALLOC, "out of host memory in rsp_mcsc_upload");
    }
    *handle = h;
    return RSP_OK;
}

int rsp_mcsc_upload(const double* x, const int32_t* p, int32_t nrow, int32_t ncol, int64_t nnz,
                    const int* devices, int ndevices, rsp_mcsc_t* handle) {
    return mcsc_upload(x, nullptr, p, nrow, ncol, nnz, devices, ndevices, handle);
}

int rsp_mcsc_upload_csc(const double* x, const int32_t* i, const int32_t* p, int32_t nrow, int32_t ncol,
                        int64_t nnz, const int* devices, int ndevices, rsp_mcsc_t* handle) {
    if (nnz > 0 && !i) return fail(RSP_ERR_BAD_ARG, "i is null (rsp_mcsc_upload takes a matrix without row indices)");
    return mcsc_upload(x, i, p, nrow, ncol, nnz, devices, ndevices, handle);
}

int rsp_mcsc_wrap_device(int nshards, const int* devices, const double* const* d_x, const int32_t* const* d_i,
                         const int32_t* const* d_p, const int32_t* shard_ncol, const int64_t* shard_nnz, int32_t nrow,
                         rsp_mcsc_t* handle) {
    if (!handle) return fail(RSP_ERR_BAD_ARG, "handle is null");
    *handle = nullptr;
    if (nshards <= 0 || !devices || !d_x || !d_p || !shard_ncol || !shard_nnz || nrow < 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_mcsc_wrap_device");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    rsp_mcsc* h = nullptr;
    int rc = RSP_OK;
    try {
        h = new rsp_mcsc();
        h->nrow = nrow;
        h->devices.assign(devices, devices + nshards);
        h->bounds.assign((size_t)nshards + 1, 0);
        h->shards.assign((size_t)nshards, nullptr);
        h->has_rows = d_i != nullptr;
        int64_t total = 0;
        for (int k = 0; k < nshards && rc == RSP_OK; ++k) {
            if (shard_ncol[k] < 0 || shard_nnz[k] < 0) rc = fail(RSP_ERR_BAD_ARG, "shard %d: negative size", k);
            total += shard_ncol[k];
            if (rc == RSP_OK && total > INT32_MAX - 65537) rc = fail(RSP_ERR_BAD_ARG, "too many columns");
            h->bounds[(size_t)k + 1] = (int32_t)total;
            if (d_i && shard_nnz[k] > 0 && !d_i[k]) h->has_rows = false;
        }
        h->ncol = (int32_t)total;
        for (int k = 0; k < nshards && rc == RSP_OK; ++k) {
            rc = rsp::csc_wrap_device(d_x[k], (d_i && h->has_rows) ? d_i[k] : nullptr, d_p[k], nrow, shard_ncol[k],
                                      shard_nnz[k], devices[k], &h->shards[(size_t)k]);
            if (rc != RSP_OK) {
                char text[400];
                snprintf(text, sizeof(text), "%s", rsp_last_error());
                rc = fail(rc, "shard %d on device %d: %s", k, devices[k], text);
            }
        }
        if (rc == RSP_OK) rc = mcsc_prepare(h);
    } catch (...) {
        rc = fail(RSP_ERR_ALLOC, "out of host memory in rsp_mcsc_wrap_device");
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc != RSP_OK) {
        char text[400];
        snprintf(text, sizeof(text), "%s", rsp_last_error());
        if (h) rsp_mcsc_free(h);
        return fail(rc, "%s", text);
    }
    *handle = h;
    return RSP_OK;
}

// RSP_GATHER_RCCL: the communicators of ONE ncclCommInitAll over the shards' devices ("single process,
// G devices") and the gathered vector on shard 0's device, made when the mode is first selected.
static int mcsc_make_rccl(rsp_mcsc* h) {
    if (!h->comms.empty()) return RSP_OK;
    const int G = (int)h->shards.size();
    for (int a = 0; a < G; ++a)
        for (int b = a + 1; b < G; ++b)
            if (h->devices[(size_t)a] == h->devices[(size_t)b])
                return fail(RSP_ERR_BAD_ARG, "the RCCL gather needs one DEVICE per shard (shards %d and %d share device %d): "
                                             "RCCL refuses two ranks of a communicator on one device", a, b, h->devices[(size_t)a]);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    int rc = RSP_OK;
    try {
        h->comms.assign((size_t)G, nullptr);
    } catch (...) {
        return fail(RSP_ERR_ALLOC, "out of host memory");
    }
    const ncclResult_t r = ncclCommInitAll(h->comms.data(), G, h->devices.data());
    if (r != ncclSuccess) {
        h->comms.clear();
        rc = fail(RSP_ERR_RCCL, "ncclCommInitAll over %d devices: %s", G, ncclGetErrorString(r));
    }
    if (rc == RSP_OK) {
        hipError_t e = hipSetDevice(h->devices[0]);
        if (e == hipSuccess) e = hipMalloc((void**)&h->d_gathered, h->ncol > 0 ? (size_t)h->ncol * 8 : 8);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            h->d_gathered = nullptr;
            mcsc_release_rccl(h);
            rc = fail(RSP_ERR_HIP, "gathered vector on device %d: %s", h->devices[0], hipGetErrorString(e));
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

int rsp_mcsc_set_gather(rsp_mcsc_t h, int mode) {
    if (!h) return fail(RSP_ERR_BAD_ARG, "null handle");
    if (mode == RSP_GATHER_RCCL) {
        if (int rc = mcsc_make_rccl(h)) return rc;
    } else if (mode == RSP_GATHER_STORES || mode == RSP_GATHER_BLIT) {
        if (!h->d_result_view) return fail(RSP_ERR_HIP, "the page-locked result vector has no device address on this system");
    } else if (mode != RSP_GATHER_D2H && mode != RSP_GATHER_NONE) {
        return fail(RSP_ERR_BAD_ARG, "unknown gather mode %d", mode);
    }
    h->gather = mode;
    return RSP_OK;
}

int rsp_mcsc_set_launch(rsp_mcsc_t h, int mode) {
    if (!h) return fail(RSP_ERR_BAD_ARG, "null handle");
    if (mode != RSP_LAUNCH_SERIAL && mode != RSP_LAUNCH_WORKERS) return fail(RSP_ERR_BAD_ARG, "unknown launch mode %d", mode);
    h->launch = mode;
    return RSP_OK;
}

int rsp_mcsc_config(rsp_mcsc_t h, int32_t* info4) {
    if (!h || !info4) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    info4[0] = h->gather;
    info4[1] = h->launch;
    info4[2] = h->workers ? h->workers->size() : 0;
    info4[3] = (int32_t)h->comms.size();
    return RSP_OK;
}

double* rsp_mcsc_result_buffer(rsp_mcsc_t h) { return h ? h->h_result : nullptr; }

static int mcsc_columns(rsp_mcsc_t h, double* sums, bool means) {
    if (!h || (h->ncol > 0 && !sums)) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    if (h->owner != getpid())
        return fail(RSP_ERR_BAD_ARG, "this handle was made by process %d and does not survive a fork (neither its threads nor the GPU context do): "
                                     "make the handle in the process that uses it", (int)h->owner);
    const int G = (int)h->shards.size();
    h->t_call_begin = now_us();
    if (h->ncol == 0) {
        h->t_call_end = h->t_call_begin;
        return RSP_OK;
    }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    ColumnCall call{h, sums, means};
    int rc = RSP_OK;
    RelaxedCapture relaxed;
    if (h->gather == RSP_GATHER_RCCL) {
        // the launches, then ONE group of sends and receives (a group is issued by one thread), then one copy home
        for (int k = 0; k < G; ++k) {
            const hipError_t e = hipSetDevice(h->devices[(size_t)k]);
            if (e != hipSuccess) shard_fail(h->st[(size_t)k], RSP_ERR_HIP, "hipSetDevice", hipGetErrorString(e));
            shard_launch(&call, k);
        }
        bool ok = true;
        for (int k = 0; k < G; ++k) ok = ok && h->st[(size_t)k].status == RSP_OK;
        ncclResult_t r = ok && G > 1 ? ncclGroupStart() : ncclSuccess;
        if (ok && G > 1 && r == ncclSuccess) {
            for (int k = 1; k < G && r == ncclSuccess; ++k) {
                const int32_t c0 = h->bounds[(size_t)k], nc = h->bounds[(size_t)k + 1] - c0;
                if (nc <= 0) continue;
                r = ncclSend(h->st[(size_t)k].view.d_out, (size_t)nc, ncclDouble, 0, h->comms[(size_t)k], h->st[(size_t)k].view.stream);
                if (r == ncclSuccess)
                    r = ncclRecv(h->d_gathered + c0, (size_t)nc, ncclDouble, k, h->comms[0], h->st[0].view.stream);
            }
            const ncclResult_t r2 = ncclGroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (r != ncclSuccess) rc = fail(RSP_ERR_RCCL, "gatherv of the slices: %s", ncclGetErrorString(r));
        hipError_t e = hipSetDevice(h->devices[0]);
        // (shard 0 may own no column at all: its stream still carries the receives)
        if (e == hipSuccess && ok && rc == RSP_OK)
            e = hipMemcpyAsync(h->h_result, h->d_gathered, (size_t)h->ncol * 8, hipMemcpyDeviceToHost, h->st[0].view.stream);
        for (int k = 0; k < G; ++k) {   // every stream drained before the call returns, whatever went wrong
            (void)hipSetDevice(h->devices[(size_t)k]);
            const hipError_t w = hipStreamSynchronize(h->st[(size_t)k].view.stream);
            if (w != hipSuccess && e == hipSuccess) e = w;
            h->st[(size_t)k].t_done = now_us();
        }
        if (e != hipSuccess && rc == RSP_OK) rc = fail(RSP_ERR_HIP, "RCCL gather: %s", hipGetErrorString(e));
        if (rc == RSP_OK && ok) {
            if (h->launch == RSP_LAUNCH_WORKERS && ensure_workers(h)) h->workers->run(shard_copy_only, &call);
            else for (int k = 0; k < G; ++k) shard_copy_only(&call, k);
        }
    } else if (h->launch == RSP_LAUNCH_WORKERS && G > 1 && ensure_workers(h)) {
        const hipError_t e = hipSetDevice(h->devices[0]);   // shard 0 is this thread's
        if (e != hipSuccess) rc = fail(RSP_ERR_HIP, "hipSetDevice(%d): %s", h->devices[0], hipGetErrorString(e));
        else h->workers->run(shard_whole, &call);
    } else {
        // every shard's kernels first (a device starts working ~3 us after the one before it), the copy commands in a
        // second pass while the kernels run, then the waits in shard order
        for (int k = 0; k < G; ++k) {
            const hipError_t e = hipSetDevice(h->devices[(size_t)k]);
            if (e != hipSuccess) shard_fail(h->st[(size_t)k], RSP_ERR_HIP, "hipSetDevice", hipGetErrorString(e));
            shard_launch(&call, k);
        }
        for (int k = 0; k < G; ++k) {
            (void)hipSetDevice(h->devices[(size_t)k]);
            shard_send(&call, k);
        }
        // the shards in the order they finish: one thread looks at the events in turn
        uint64_t pending = 0;   // (more than 64 shards: the tail is waited for in order afterwards)
        for (int k = 0; k < G && k < 64; ++k) pending |= (uint64_t)1 << k;
        int current = h->devices[(size_t)G - 1];
        while (pending) {
            for (int k = 0; k < G && k < 64; ++k) {
                if (!((pending >> k) & 1)) continue;
                if (h->devices[(size_t)k] != current) {
                    current = h->devices[(size_t)k];
                    (void)hipSetDevice(current);
                }
                if (shard_poll(&call, k)) pending &= ~((uint64_t)1 << k);
            }
            if (pending) cpu_relax();
        }
        for (int k = 64; k < G; ++k) {
            (void)hipSetDevice(h->devices[(size_t)k]);
            while (!shard_poll(&call, k)) cpu_relax();
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    h->t_call_end = now_us();
    if (rc != RSP_OK) return rc;
    for (int k = 0; k < G; ++k)
        if (h->st[(size_t)k].status != RSP_OK)
            return fail(h->st[(size_t)k].status, "shard %d on device %d: %s", k, h->devices[(size_t)k], h->st[(size_t)k].message);
    return RSP_OK;
}

int rsp_mcsc_column_sums(rsp_mcsc_t h, double* sums) { return mcsc_columns(h, sums, false); }
int rsp_mcsc_column_means(rsp_mcsc_t h, double* means) { return mcsc_columns(h, means, true); }

// Measurement: the host clock of the LAST column-sum call on this handle, microseconds from the call's entry:
// us[0] = the whole call; then per shard { enqueue begun, enqueue returned, stream drained, slice copied out }.
int rsp_mcsc_last_call_stamps(rsp_mcsc_t h, double* us, int capacity) {
    if (!h || !us) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    const int G = (int)h->shards.size();
    if (capacity < 1 + 4 * G) return fail(RSP_ERR_BAD_ARG, "room for %d values needed", 1 + 4 * G);
    us[0] = h->t_call_end - h->t_call_begin;
    for (int k = 0; k < G; ++k) {
        const ShardState& s = h->st[(size_t)k];
        us[1 + 4 * k + 0] = s.t_begin - h->t_call_begin;
        us[1 + 4 * k + 1] = s.t_enqueued - h->t_call_begin;
        us[1 + 4 * k + 2] = s.t_done - h->t_call_begin;
        us[1 + 4 * k + 3] = s.t_copied - h->t_call_begin;
    }
    return RSP_OK;
}

// Measurement: mean device time of ONE shard's column-sum launches alone (HIP events on the shard's stream, `reps`
// calls back to back after one untimed call) -- what the call's wall time is compared with.
int rsp_mcsc_shard_kernel_ms(rsp_mcsc_t h, int32_t shard, int reps, float* ms) {
    if (!h || !ms || reps <= 0) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_mcsc_shard_kernel_ms");
    if (shard < 0 || shard >= (int32_t)h->shards.size()) return fail(RSP_ERR_BAD_ARG, "shard %d out of range", shard);
    *ms = 0.0f;
    if (h->bounds[(size_t)shard + 1] == h->bounds[(size_t)shard]) return RSP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    const ShardState& s = h->st[(size_t)shard];
    hipEvent_t a = nullptr, b = nullptr;
    int rc = RSP_OK;
    hipError_t e = hipSetDevice(h->devices[(size_t)shard]);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) rc = rsp::csc_enqueue_columns(h->shards[(size_t)shard], false, nullptr);
    if (e == hipSuccess && rc == RSP_OK) e = hipEventRecord(a, s.view.stream);
    for (int r = 0; r < reps && e == hipSuccess && rc == RSP_OK; ++r) rc = rsp::csc_enqueue_columns(h->shards[(size_t)shard], false, nullptr);
    if (e == hipSuccess && rc == RSP_OK) e = hipEventRecord(b, s.view.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s.view.stream);
    float total = 0.0f;
    if (e == hipSuccess && rc == RSP_OK) e = hipEventElapsedTime(&total, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc != RSP_OK) return rc;
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "timing shard %d: %s", shard, hipGetErrorString(e));
    *ms = total / (float)reps;
    return RSP_OK;
}

// Matrix::rowSums / rowMeans (reference RcppSparse.h:138-156) of the resident shards.  A row's sum is its shards' partial
// sums added in SHARD order (= column order, the order the reference's scatter loop meets the entries in).
// Round 6: the add happens ON THE DEVICES -- the single-process form of rsp_comm_reduce_rows, same sum term for term:
//   1. every shard sums the rows of its own columns into a vector of nrow doubles in its HBM;
//   2. the rows are cut into G slices; device r copies slice r of every other shard's vector to itself (hipMemcpyPeerAsync
//      on ITS stream behind an event of the source: every pair of devices its own xGMI link), adds the G pieces in shard
//      order with one kernel (rows_add_partials_kernel) and
//   3. writes its reduced slice straight into a page-locked host vector over its own host link; the handle's worker
//      threads copy the slices out into the caller's vector side by side.
// 8 shards x 1e7 rows: 70 MB in and out of every device over xGMI and 10 MB per host link, instead of 80 MB per host link
// and 640 MB added by the host.  The buffers (per device: nrow + (G + 1) * nrow / G doubles) are made at the first call and
// kept; if they cannot be had -- or RSP_MCSC_ROWS=host -- the host-side add below does the call, with the same bits.
static int64_t reduce_slice_len(int nranks, int32_t nrow);

static bool mcsc_rows_prepare(rsp_mcsc* h) {
    if (h->rows_ready) return true;
    if (h->rows_failed) return false;
    static const bool host_only = [] {
        const char* e = getenv("RSP_MCSC_ROWS");
        return e && !strcmp(e, "host");
    }();
    const size_t G = h->shards.size();
    const int64_t len = reduce_slice_len((int)G, h->nrow);
    bool ok = !host_only;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    if (ok) {
        try {
            h->rows.assign(G, rsp_mcsc::RowShard());
        } catch (...) {
            ok = false;
        }
    }
    if (ok) {   // (without it the reduced slices are copied home from d_reduced: slower into pageable memory, not wrong)
        if (hipHostMalloc((void**)&h->h_rows, (size_t)(h->nrow > 0 ? h->nrow : 1) * 8, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) {
            void* dv = nullptr;
            if (hipHostGetDevicePointer(&dv, h->h_rows, 0) == hipSuccess) h->d_rows_view = (double*)dv;
            else (void)hipGetLastError();
        } else {
            (void)hipGetLastError();
            h->h_rows = nullptr;
        }
    }
    for (size_t k = 0; k < G && ok; ++k) {
        rsp_mcsc::RowShard& r = h->rows[k];
        hipError_t e = hipSetDevice(h->devices[k]);
        if (e == hipSuccess) e = hipMalloc((void**)&r.d_partial, (size_t)h->nrow * 8);
        if (e == hipSuccess && G > 1) e = hipMalloc((void**)&r.d_incoming, G * (size_t)len * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&r.d_reduced, (size_t)len * 8);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.computed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
        }
    }
    // the slices travel device to device: let every device reach the others directly where the hardware can (without it
    // the runtime stages a peer copy through the host); what cannot be enabled simply stays staged
    for (size_t a = 0; a < G && ok; ++a)
        for (size_t b = 0; b < G; ++b) {
            if (h->devices[a] == h->devices[b]) continue;
            bool seen = false;   // (once per ordered pair of DEVICES, however many shards share them)
            for (size_t a2 = 0; a2 <= a && !seen; ++a2)
                for (size_t b2 = 0; b2 < (a2 == a ? b : G) && !seen; ++b2)
                    seen = h->devices[a2] == h->devices[a] && h->devices[b2] == h->devices[b];
            if (seen) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, h->devices[a], h->devices[b]) == hipSuccess && can &&
                hipSetDevice(h->devices[a]) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(h->devices[b], 0);
                if (e != hipSuccess) (void)hipGetLastError();   // (already enabled, or not to be had: either is fine)
            } else {
                (void)hipGetLastError();
            }
        }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (!ok) {   // (what was made is released with the handle; the host-side add takes over for good)
        h->rows_failed = true;
        return false;
    }
    h->rows_ready = true;
    return true;
}

namespace {
struct RowCall {
    rsp_mcsc* h;
    double* out;
    int64_t len;
};
// step 3 for shard k: wait for its reduced slice, copy it home (pageable destination: the runtime stages it)
void rows_copy_home(void* ctx, int k) noexcept {
    RowCall* c = (RowCall*)ctx;
    rsp_mcsc* h = c->h;
    ShardState& s = h->st[(size_t)k];
    const int64_t first = (int64_t)k * c->len < h->nrow ? (int64_t)k * c->len : (int64_t)h->nrow;
    const int64_t next = (int64_t)(k + 1) * c->len < h->nrow ? (int64_t)(k + 1) * c->len : (int64_t)h->nrow;
    hipError_t e = hipEventSynchronize(h->rows[(size_t)k].done);
    if (e == hipSuccess && next > first) {
        if (h->d_rows_view) memcpy(c->out + first, h->h_rows + first, (size_t)(next - first) * 8);   // (the kernel wrote it there)
        else e = hipMemcpy(c->out + first, h->rows[(size_t)k].d_reduced, (size_t)(next - first) * 8, hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) shard_fail(s, RSP_ERR_HIP, "row sums: the reduced slice's way home", hipGetErrorString(e));
}
}  // namespace

static int mcsc_rows_on_devices(rsp_mcsc* h, double* out, bool means) {
    const int G = (int)h->shards.size();
    const int64_t len = reduce_slice_len(G, h->nrow);
    auto first = [&](int k) { const int64_t f = (int64_t)k * len; return f < h->nrow ? f : (int64_t)h->nrow; };
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    RelaxedCapture relaxed;
    for (int k = 0; k < G; ++k) h->st[(size_t)k].status = RSP_OK;
    // 1. every shard's partial vector
    for (int k = 0; k < G; ++k) {
        ShardState& s = h->st[(size_t)k];
        hipError_t e = hipSetDevice(h->devices[(size_t)k]);
        if (e == hipSuccess) {
            if (h->bounds[(size_t)k + 1] > h->bounds[(size_t)k]) {
                const int rc = rsp::csc_enqueue_rows(h->shards[(size_t)k], h->rows[(size_t)k].d_partial);
                if (rc != RSP_OK) shard_fail(s, rc, "row sums", rsp_last_error());
            } else {   // a shard without columns adds nothing
                e = hipMemsetAsync(h->rows[(size_t)k].d_partial, 0, (size_t)h->nrow * 8, s.view.stream);
            }
        }
        if (e == hipSuccess) e = hipEventRecord(h->rows[(size_t)k].computed, s.view.stream);
        if (e != hipSuccess) shard_fail(s, RSP_ERR_HIP, "row sums: enqueue", hipGetErrorString(e));
    }
    bool ok = true;
    for (int k = 0; k < G; ++k) ok = ok && h->st[(size_t)k].status == RSP_OK;
    // 2. slice r of every vector to device r, added there in shard order
    for (int r = 0; r < G && ok; ++r) {
        ShardState& s = h->st[(size_t)r];
        rsp_mcsc::RowShard& me = h->rows[(size_t)r];
        const int64_t cnt = first(r + 1) - first(r);
        hipError_t e = hipSetDevice(h->devices[(size_t)r]);
        for (int k = 0; k < G && e == hipSuccess && cnt > 0; ++k) {
            if (k == r) continue;
            e = hipStreamWaitEvent(s.view.stream, h->rows[(size_t)k].computed, 0);
            const double* src = h->rows[(size_t)k].d_partial + first(r);
            double* dst = me.d_incoming + (size_t)k * (size_t)len;
            if (e == hipSuccess)
                e = h->devices[(size_t)k] == h->devices[(size_t)r]
                        ? hipMemcpyAsync(dst, src, (size_t)cnt * 8, hipMemcpyDeviceToDevice, s.view.stream)
                        : hipMemcpyPeerAsync(dst, h->devices[(size_t)r], src, h->devices[(size_t)k], (size_t)cnt * 8, s.view.stream);
        }
        if (e == hipSuccess && cnt > 0)
            // (the reduced slice: consecutive doubles, written by neighbouring lanes -- straight into the page-locked host vector
            // where the device can address it, over the device's own host link)
            e = rsp::launch_add_partials(G > 1 ? me.d_incoming : me.d_partial, G, len, me.d_partial + first(r), r, cnt,
                                         h->d_rows_view ? h->d_rows_view + first(r) : me.d_reduced, means ? (double)h->ncol : 1.0,
                                         means, s.view.stream);
        if (e == hipSuccess) e = hipEventRecord(me.done, s.view.stream);
        if (e != hipSuccess) {
            shard_fail(s, RSP_ERR_HIP, "row sums: reduce", hipGetErrorString(e));
            ok = false;
        }
    }
    // 3. the reduced slices home, side by side where the handle has its workers
    if (ok) {
        RowCall call{h, out, len};
        if (G > 1 && ensure_workers(h)) {
            (void)hipSetDevice(h->devices[0]);
            h->workers->run(rows_copy_home, &call);
        } else {
            for (int k = 0; k < G; ++k) {
                (void)hipSetDevice(h->devices[(size_t)k]);
                rows_copy_home(&call, k);
            }
        }
    }
    for (int k = 0; k < G; ++k) {   // nothing of a failed call stays in flight
        if (ok) break;
        (void)hipSetDevice(h->devices[(size_t)k]);
        (void)hipStreamSynchronize(h->st[(size_t)k].view.stream);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    for (int k = 0; k < G; ++k)
        if (h->st[(size_t)k].status != RSP_OK)
            return fail(h->st[(size_t)k].status, "shard %d on device %d: %s", k, h->devices[(size_t)k], h->st[(size_t)k].message);
    return RSP_OK;
}

// The host-side add (until round 6 the only form; now the fall-back): every shard's partial vector comes back over its own
// device's link into a host vector, and the host adds the vectors in shard order.
static int mcsc_rows(rsp_mcsc_t h, double* out, bool means) try {
    if (!h || (h->nrow > 0 && !out)) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    if (!h->has_rows)
        return fail(RSP_ERR_BAD_ARG, "this handle was uploaded without i[]: use rsp_mcsc_upload_csc for the row-wise entries");
    if (h->owner != getpid())
        return fail(RSP_ERR_BAD_ARG, "this handle was made by process %d and does not survive a fork: make the handle in the process that uses it",
                    (int)h->owner);
    const int G = (int)h->shards.size();
    const size_t nrow = (size_t)h->nrow;
    if (nrow == 0) return RSP_OK;
    if (mcsc_rows_prepare(h)) return mcsc_rows_on_devices(h, out, means);
    // shards x nrow doubles for the duration of this call only (8 shards x 1e7 rows: 640 MB), so that the handle
    // holds no host memory between calls and two threads asking the same handle for row sums do not share a buffer
    std::vector<double> partial((size_t)G * nrow);
    std::vector<int> status((size_t)G, RSP_OK);
    std::vector<std::string> message((size_t)G);
    auto work = [&](int k) noexcept {
        double* mine = partial.data() + (size_t)k * nrow;
        if (h->bounds[k + 1] == h->bounds[k]) {   // a shard without columns adds nothing
            for (size_t r = 0; r < nrow; ++r) mine[r] = 0.0;
            return;
        }
        status[k] = rsp_csc_row_sums(h->shards[k], mine);
        if (status[k] != RSP_OK) {
            try {
                message[k] = rsp_last_error();
            } catch (...) {
            }
        }
    };
    if (!run_shards(G, work)) return fail(RSP_ERR_ALLOC, "out of host memory or threads while summing the shards");
    for (int k = 0; k < G; ++k)
        if (status[k] != RSP_OK) return fail(status[k], "shard %d: %s", k, message[k].c_str());
    // the add: rows are independent, so ranges of rows go to host threads; per row the order stays shard 0, 1, ...
    const double divisor = (double)h->ncol;
    const int nthreads = nrow >= ((size_t)1 << 18) ? 8 : 1;
    auto add = [&](int t) noexcept {
        const size_t r0 = nrow * (size_t)t / (size_t)nthreads, r1 = nrow * (size_t)(t + 1) / (size_t)nthreads;
        for (size_t r = r0; r < r1; ++r) {
            double v = partial[r];
            for (int k = 1; k < G; ++k) v += partial[(size_t)k * nrow + r];
            v += 0.0;
            out[r] = means ? v / divisor : v;   // RcppSparse.h:153-154
        }
    };
    if (nthreads == 1 || !run_shards(nthreads, add))
        for (int t = 0; t < nthreads; ++t) add(t);   // (no threads to be had: the same ranges on this one)
    return RSP_OK;
} catch (...) {
    return fail(RSP_ERR_ALLOC, "out of host memory in rsp_mcsc_row_sums");
}

int rsp_mcsc_row_sums(rsp_mcsc_t h, double* sums) { return mcsc_rows(h, sums, false); }

int rsp_mcsc_dims(rsp_mcsc_t h, int32_t* nrow, int32_t* ncol, int32_t* nshards) {
    if (!h) return fail(RSP_ERR_BAD_ARG, "null handle");
    if (nrow) *nrow = h->nrow;
    if (ncol) *ncol = h->ncol;
    if (nshards) *nshards = (int32_t)h->shards.size();
    return RSP_OK;
}

int rsp_mcsc_shard_info(rsp_mcsc_t h, int32_t shard, int32_t* info4) {
    if (!h || !info4) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    if (shard < 0 || shard >= (int32_t)h->shards.size()) return fail(RSP_ERR_BAD_ARG, "shard %d out of range", shard);
    info4[0] = h->bounds[shard];
    info4[1] = h->bounds[shard + 1];
    info4[2] = rsp_csc_column_form(h->shards[shard]);
    int64_t nnz = 0;
    (void)rsp_csc_dims(h->shards[shard], nullptr, nullptr, &nnz);
    info4[3] = (int32_t)nnz;
    return RSP_OK;
}
int rsp_mcsc_row_means(rsp_mcsc_t h, double* means) { return mcsc_rows(h, means, true); }

// Which RCCL this process really runs: the version the loaded library reports and the file it was mapped from.  The
// library is linked against librccl.so by soname; which file answers is the loader's choice (a process that imported torch
// first gets torch's bundled copy) -- a multi-GPU number should say which one produced it.
int rsp_rccl_info(int* version, char* library_path, size_t capacity) {
    if (version) {
        int v = 0;
        const ncclResult_t r = ncclGetVersion(&v);
        if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGetVersion: %s", ncclGetErrorString(r));
        *version = v;
    }
    if (library_path && capacity > 0) {
        library_path[0] = '\0';
        FILE* f = fopen("/proc/self/maps", "r");
        if (f) {
            char line[1024];
            while (fgets(line, sizeof(line), f)) {
                const char* hit = strstr(line, "librccl");
                if (!hit) continue;
                const char* path = strchr(line, '/');
                if (!path) continue;
                size_t n = strcspn(path, "\n");
                if (n >= capacity) n = capacity - 1;
                memcpy(library_path, path, n);
                library_path[n] = '\0';
                break;
            }
            fclose(f);
        }
    }
    return RSP_OK;
}

int rsp_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return fail(RSP_ERR_BAD_ARG, "id_bytes is null");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id_bytes, &id, sizeof(id));
    return RSP_OK;
}

int rsp_comm_init(const void* id_bytes, int nranks, int rank, int device, rsp_comm_t* comm) {
    if (!id_bytes || !comm || nranks <= 0 || rank < 0 || rank >= nranks)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_comm_init");
    *comm = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    rsp_comm* c = new (std::nothrow) rsp_comm();
    if (!c) return fail(RSP_ERR_ALLOC, "out of host memory");
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(RSP_ERR_RCCL, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    *comm = c;
    return RSP_OK;
}

int rsp_comm_gatherv(rsp_comm_t c, const double* d_send, int64_t send_count, double* d_recv,
                     const int64_t* counts, const int64_t* displs, int root, void* stream) {
    if (!c || root < 0 || root >= c->nranks || send_count < 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_comm_gatherv");
    hipStream_t s = (hipStream_t)stream;
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGroupStart: %s", ncclGetErrorString(r));
    if (c->rank == root) {
        if (!counts || !displs || !d_recv) {
            (void)ncclGroupEnd();
            return fail(RSP_ERR_BAD_ARG, "root needs d_recv, counts and displs");
        }
        for (int k = 0; k < c->nranks && r == ncclSuccess; ++k) {
            if (k == root) {
                // own slice: device-to-device copy on the same stream (skipped when in place)
                if (counts[k] > 0 && d_send != d_recv + displs[k]) {
                    hipError_t e = hipMemcpyAsync(d_recv + displs[k], d_send, (size_t)counts[k] * 8,
                                                  hipMemcpyDeviceToDevice, s);
                    if (e != hipSuccess) {
                        (void)ncclGroupEnd();
                        return fail(RSP_ERR_HIP, "root copy: %s", hipGetErrorString(e));
                    }
                }
            } else if (counts[k] > 0) {
                r = ncclRecv(d_recv + displs[k], (size_t)counts[k], ncclDouble, k, c->comm, s);
            }
        }
    } else if (send_count > 0) {
        r = ncclSend(d_send, (size_t)send_count, ncclDouble, root, c->comm, s);
    }
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclSend/Recv: %s", ncclGetErrorString(r));
    if (r2 != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGroupEnd: %s", ncclGetErrorString(r2));
    return RSP_OK;
}

// ---- rowSums over column-range shards: reduce of f64[nrow] in RANK ORDER ------------------------
// Every rank holds the partial row sums of its own columns (reference RcppSparse.h:138-144 restricted to
// the shard).  The full sum of a row is its partials added in rank order = column order, the order the
// reference's scatter loop meets the entries in; ncclReduce would add them in whatever order RCCL's
// ring or tree visits the ranks, which depends on topology and message size.  So the reduce is built from
// point-to-point transfers and one add kernel:
//   1. the rows are cut into nranks slices; rank r receives slice r of every other rank's partial vector
//      (an all-to-all of nrow / nranks doubles per pair, every pair on its own xGMI link),
//   2. adds the nranks pieces of its slice in rank order (rows_add_partials_kernel),
//   3. and the reduced slices are gathered to the root (rsp_comm_gatherv).
// Over the root's links travel (nranks - 1) / nranks * nrow doubles instead of (nranks - 1) * nrow for
// "gather every partial vector to the root, add there" -- 70 MB instead of 560 MB at 8 ranks and 1e7 rows --
// and the result has the same bits as that simpler scheme and as a single process adding the shards in order.
static int64_t reduce_slice_len(int nranks, int32_t nrow) {
    int64_t len = ((int64_t)nrow + nranks - 1) / nranks;
    return (len + 1) & ~(int64_t)1;   // whole 16-byte pairs: every slice starts 16-byte aligned
}

size_t rsp_comm_reduce_rows_workspace_bytes(int nranks, int32_t nrow) {
    if (nranks <= 0 || nrow < 0) return 0;
    // nranks incoming pieces (the own slot stays unused) + the reduced slice
    return ((size_t)(nranks + 1) * (size_t)reduce_slice_len(nranks, nrow) * 8 + 255) & ~(size_t)255;
}

int rsp_add_partials_device(const double* d_parts, int32_t nparts, int64_t stride, int64_t n,
                            int32_t ncol_for_means, double* d_out, void* stream) {
    if (nparts <= 0 || n < 0 || stride < n || (n > 0 && (!d_parts || !d_out)) || ncol_for_means < 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_add_partials_device");
    hipError_t e = rsp::launch_add_partials(d_parts, nparts, stride, nullptr, -1, n, d_out,
                                            ncol_for_means > 0 ? (double)ncol_for_means : 1.0, ncol_for_means > 0,
                                            (hipStream_t)stream);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "rows_add_partials_kernel: %s", hipGetErrorString(e));
    return RSP_OK;
}

int rsp_comm_reduce_rows(rsp_comm_t c, const double* d_partial, int32_t nrow, int32_t ncol_for_means,
                         double* d_result, void* d_workspace, size_t workspace_bytes, int root, void* stream) {
    if (!c || root < 0 || root >= c->nranks || nrow < 0 || ncol_for_means < 0 || (nrow > 0 && !d_partial))
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_comm_reduce_rows");
    if (c->rank == root && nrow > 0 && !d_result) return fail(RSP_ERR_BAD_ARG, "root needs d_result");
    if (nrow == 0) return RSP_OK;
    const int G = c->nranks, me = c->rank;
    if (!d_workspace || workspace_bytes < rsp_comm_reduce_rows_workspace_bytes(G, nrow))
        return fail(RSP_ERR_WORKSPACE, "workspace too small: need %zu bytes", rsp_comm_reduce_rows_workspace_bytes(G, nrow));
    hipStream_t s = (hipStream_t)stream;
    const int64_t len = reduce_slice_len(G, nrow);
    auto first = [&](int k) { const int64_t f = (int64_t)k * len; return f < nrow ? f : (int64_t)nrow; };
    auto count = [&](int k) { return first(k + 1) - first(k); };
    double* incoming = (double*)d_workspace;           // piece k at incoming + k * len
    double* reduced = incoming + (size_t)G * (size_t)len;
    // 1. all-to-all of slices
    if (G > 1) {
        ncclResult_t r = ncclGroupStart();
        if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGroupStart: %s", ncclGetErrorString(r));
        for (int k = 0; k < G && r == ncclSuccess; ++k) {
            if (k == me) continue;
            if (count(k) > 0) r = ncclSend(d_partial + first(k), (size_t)count(k), ncclDouble, k, c->comm, s);
            if (r == ncclSuccess && count(me) > 0)
                r = ncclRecv(incoming + (size_t)k * (size_t)len, (size_t)count(me), ncclDouble, k, c->comm, s);
        }
        ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclSend/Recv: %s", ncclGetErrorString(r));
        if (r2 != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGroupEnd: %s", ncclGetErrorString(r2));
    }
    // 2. this rank's slice: the G pieces in rank order (its own piece straight from d_partial)
    hipError_t e = rsp::launch_add_partials(incoming, G, len, d_partial + first(me), me, count(me), reduced,
                                            ncol_for_means > 0 ? (double)ncol_for_means : 1.0, ncol_for_means > 0, s);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "rows_add_partials_kernel: %s", hipGetErrorString(e));
    // 3. reduced slices to the root
    int64_t counts[64], displs[64];
    std::vector<int64_t> big;
    int64_t *pc = counts, *pd = displs;
    if (G > 64) {
        try {
            big.resize((size_t)G * 2);
        } catch (...) {
            return fail(RSP_ERR_ALLOC, "out of host memory");
        }
        pc = big.data();
        pd = big.data() + G;
    }
    for (int k = 0; k < G; ++k) {
        pc[k] = count(k);
        pd[k] = first(k);
    }
    return rsp_comm_gatherv(c, reduced, count(me), d_result, pc, pd, root, stream);
}

// ---- direct-write gather: the root's result buffer mapped into every rank's address space ------------------
// The gatherv above moves every rank's slice with a send / receive pair.  On one node the root can instead hand its
// result buffer to the other rank processes (hipIpcGetMemHandle / hipIpcOpenMemHandle): a rank's column-sum kernels
// then take `mapped + displacement` as their output pointer and the "gather" is their own result stores, travelling
// over xGMI as they are made; what is left of the exchange is one fence (every rank's kernels done, then a barrier
// between the rank processes).  SURVEY.md section 5 calls this a legitimate comparator beside RCCL; bench.py reports
// it as `direct_gather`, never as `value`.
static_assert(RSP_IPC_HANDLE_BYTES >= sizeof(hipIpcMemHandle_t), "ipc handle size");

int rsp_shared_result_alloc(size_t bytes, void** d_ptr, void* handle_bytes) {
    if (!d_ptr || !handle_bytes || bytes == 0) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_shared_result_alloc");
    *d_ptr = nullptr;
    void* q = nullptr;
    // fine-grained: stores arriving from other devices are coherent with this device's later reads without relying on
    // where the home L2 keeps its lines (the buffer is a few MB of results, not a stream)
    hipError_t e = hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "allocating the shared result failed: %s", hipGetErrorString(e));
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, q);
    if (e != hipSuccess) {
        (void)hipFree(q);
        return fail(RSP_ERR_HIP, "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set before the runtime started?)",
                    hipGetErrorString(e));
    }
    memset(handle_bytes, 0, RSP_IPC_HANDLE_BYTES);
    memcpy(handle_bytes, &h, sizeof(h));
    *d_ptr = q;
    return RSP_OK;
}

int rsp_shared_result_open(const void* handle_bytes, void** d_ptr) {
    if (!d_ptr || !handle_bytes) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_shared_result_open");
    *d_ptr = nullptr;
    hipIpcMemHandle_t h;
    memcpy(&h, handle_bytes, sizeof(h));
    void* q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *d_ptr = q;
    return RSP_OK;
}

int rsp_shared_result_read(const void* d_ptr, size_t offset_bytes, void* host, size_t bytes, void* stream) {
    if (!d_ptr || (bytes > 0 && !host)) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_shared_result_read");
    hipError_t e = hipMemcpyAsync(host, (const char*)d_ptr + offset_bytes, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "reading the shared result failed: %s", hipGetErrorString(e));
    return RSP_OK;
}

int rsp_shared_result_close(void* d_ptr, int owner) {
    if (!d_ptr) return RSP_OK;
    hipError_t e = owner ? hipFree(d_ptr) : hipIpcCloseMemHandle(d_ptr);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "releasing the shared result failed: %s", hipGetErrorString(e));
    return RSP_OK;
}

// ---- host-memory gather: one page-locked vector shared by the rank processes -------------------------------------
// SURVEY.md section 5's other comparator: "per-GPU D2H into disjoint slices of one pinned buffer".  The vector lives in
// POSIX shared memory; every rank maps it and page-locks ITS mapping (hipHostRegister), so its copy engine writes the
// rank's slice straight into memory the root process reads -- over the rank's own host link, no xGMI hop, no RCCL.
// After the ranks' stream waits and one host barrier the root holds the complete vector in HOST memory (where an R
// NumericVector has to end up anyway).
int rsp_shared_host_open(const char* name, size_t bytes, int create, void** host_ptr) {
    if (!name || name[0] != '/' || !host_ptr || bytes == 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_shared_host_open (name must start with '/')");
    *host_ptr = nullptr;
    const size_t mapped = (bytes + 4095) & ~(size_t)4095;
    int fd = -1;
    if (create) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)mapped) != 0) {
            close(fd);
            (void)shm_unlink(name);
            fd = -1;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= mapped) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    if (fd < 0) return fail(RSP_ERR_ALLOC, "shared memory %s could not be %s", name, create ? "created" : "opened");
    void* m = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(RSP_ERR_ALLOC, "mmap of %s failed", name);
    const hipError_t e = hipHostRegister(m, mapped, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        munmap(m, mapped);
        if (create) (void)shm_unlink(name);
        return fail(RSP_ERR_HIP, "hipHostRegister of the shared vector: %s", hipGetErrorString(e));
    }
    *host_ptr = m;
    return RSP_OK;
}

int rsp_shared_host_close(void* host_ptr, size_t bytes, const char* unlink_name) {
    if (!host_ptr) return RSP_OK;
    const size_t mapped = (bytes + 4095) & ~(size_t)4095;
    const hipError_t e = hipHostUnregister(host_ptr);
    if (e != hipSuccess) (void)hipGetLastError();
    munmap(host_ptr, mapped);
    if (unlink_name) (void)shm_unlink(unlink_name);
    return RSP_OK;
}

// enqueue only: n doubles from device memory into (page-locked) host memory on `stream`
int rsp_copy_to_host_async(const double* d_src, double* host_dst, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!d_src || !host_dst))) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_copy_to_host_async");
    if (n == 0) return RSP_OK;
    // page-locked memory the device can address: the copy KERNEL (it starts a few microseconds behind the kernels in front
    // of it; the runtime's copy command needs ~20 us before its first byte moves -- profiles/r06_mcsc_overhead.md section 3)
    void* dv = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dv, host_dst, 0);
    if (e == hipSuccess && dv) {
        e = rsp::launch_copy_f64(d_src, (double*)dv, n, (hipStream_t)stream);
    } else {
        (void)hipGetLastError();
        e = hipMemcpyAsync(host_dst, d_src, (size_t)n * 8, hipMemcpyDeviceToHost, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "D2H copy: %s", hipGetErrorString(e));
    return RSP_OK;
}

// hipDeviceCanAccessPeer, for the probe bench.py runs in a child process before it lets kernels store across devices
int rsp_device_can_access_peer(int device, int peer, int* can) {
    if (!can) return fail(RSP_ERR_BAD_ARG, "can is null");
    *can = 0;
    if (device == peer) {
        *can = 1;
        return RSP_OK;
    }
    const hipError_t e = hipDeviceCanAccessPeer(can, device, peer);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RSP_ERR_HIP, "hipDeviceCanAccessPeer(%d, %d): %s", device, peer, hipGetErrorString(e));
    }
    return RSP_OK;
}

// A barrier between the rank PROCESSES of one node through a page of POSIX shared memory: sense-reversing, spinning
// on the host (the ranks are about to consume each other's results: they have nothing else to do), a microsecond or
// two per crossing, with a timeout instead of a hang when a rank has died.
struct rsp_host_barrier {
    struct Page {
        std::atomic<int32_t> arrived;
        std::atomic<int32_t> generation;
    };
    Page* page;
    int nranks, rank;
    std::string name;
};

int rsp_host_barrier_create(const char* name, int nranks, int rank, rsp_host_barrier_t* out) {
    if (!name || name[0] != '/' || !out || nranks <= 0 || rank < 0 || rank >= nranks)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_host_barrier_create (name must start with '/')");
    *out = nullptr;
    int fd = -1;
    if (rank == 0) {   // rank 0 creates and sizes the page; the others wait for it to appear with its size
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, 4096) != 0) {
            close(fd);
            (void)shm_unlink(name);
            fd = -1;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && st.st_size >= 4096) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    if (fd < 0) return fail(RSP_ERR_ALLOC, "shared memory %s could not be %s", name, rank == 0 ? "created" : "opened");
    void* m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(RSP_ERR_ALLOC, "mmap of %s failed", name);
    rsp_host_barrier* b = new (std::nothrow) rsp_host_barrier();
    if (!b) {
        munmap(m, 4096);
        return fail(RSP_ERR_ALLOC, "out of host memory");
    }
    b->page = (rsp_host_barrier::Page*)m;   // (a fresh page of shared memory is zero-filled: both counters start at 0)
    b->nranks = nranks;
    b->rank = rank;
    b->name = name;
    *out = b;
    return RSP_OK;
}

int rsp_host_barrier_wait(rsp_host_barrier_t b, double timeout_seconds) {
    if (!b) return fail(RSP_ERR_BAD_ARG, "null barrier");
    const int32_t gen = b->page->generation.load(std::memory_order_acquire);
    if (b->page->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == b->nranks) {
        b->page->arrived.store(0, std::memory_order_relaxed);
        b->page->generation.store(gen + 1, std::memory_order_release);   // releases the others
        return RSP_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (b->page->generation.load(std::memory_order_acquire) == gen) {
        if ((++spins & 1023u) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds)
            return fail(RSP_ERR_RCCL, "host barrier timed out after %.1f s (a rank is missing)", timeout_seconds);
    }
    return RSP_OK;
}

int rsp_host_barrier_destroy(rsp_host_barrier_t b) {
    if (!b) return RSP_OK;
    munmap((void*)b->page, 4096);
    if (b->rank == 0) (void)shm_unlink(b->name.c_str());
    delete b;
    return RSP_OK;
}

int rsp_comm_destroy(rsp_comm_t c) {
    if (!c) return RSP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(c->device);
    ncclResult_t r = ncclCommDestroy(c->comm);
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    delete c;
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclCommDestroy: %s", ncclGetErrorString(r));
    return RSP_OK;
}

}  // extern "C"

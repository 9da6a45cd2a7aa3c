// capi.hip -- the extern "C" shim declared in include/rcppsparse_hip.h.
//
// Sits directly under the Rcpp-side columnSums (reference src/example.cpp:26-32):
// the caller hands over REAL(x), INTEGER(p), ncol, nnz and a pre-allocated
// output; nothing here knows about R, Rcpp or torch.  No CPU fallback: without
// a HIP device every compute entry fails with RSP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include <atomic>
#include <mutex>
#include <string>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unistd.h>

#include "../../include/rcppsparse_hip.h"
#include "colsums_kernels.h"
#include "inspect.hpp"

#ifdef RSP_STAMPS
namespace rsp { hipError_t read_stamps(unsigned long long* host, size_t n); }
#endif

namespace {

thread_local char g_err[512] = "";
// tuning knobs are read by the worker threads of the multi-GPU host entries: atomics
std::atomic<int> g_chunk_rows_override{0};   // rsp_set_tuning / RSP_CHUNK_ROWS
std::atomic<int> g_variant{-1};              // rsp_set_experiment / RSP_VARIANT (-1 = read the env)
std::atomic<int> g_crossprod_exact{-1};      // rsp_set_crossprod_exact / RSP_CROSSPROD_EXACT (-1 = read the env)
std::atomic<int> g_taper_permille{-1};       // rsp_set_taper / RSP_TAPER (-1 = env, else the default)
std::atomic<int> g_taper_rows{-1};

int env_int(const char* name) {
    const char* s = getenv(name);
    const int v = s ? atoi(s) : 0;
    return v < 0 ? 0 : v;
}

}  // namespace

namespace rsp {
// records the message for rsp_last_error() on this thread and returns `code`
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace rsp

namespace {
using rsp::fail;

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(RSP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

std::atomic<int> g_lean{-1};   // rsp_set_lean: 0 / 1 / 2; -1 = RSP_LEAN from the environment, else 1
// 0: plans never take the lean form; 1: where it is the faster one (every column <= 64 entries AND a mean of at most
// kLeanMaxMeanLen = 60: with longer columns a chunk of 16 rows holds too few of them for its 64 lanes -- columns of exactly
// 64: 33.0 us against 19.6 snapped; means of 20..56 at 1e7 / 1e8 entries: lean within 0.85..1.05 of snapped, with the
// reference's bits; profiles/r04_form_edges.json); 2: wherever it applies at all (tests, edge sweeps)
int lean_setting() {
    int v = g_lean.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = [] {
            const char* s = getenv("RSP_LEAN");
            const int e = s ? atoi(s) : 1;
            return e <= 0 ? 0 : (e >= 2 ? 2 : 1);
        }();
        v = env;
    }
    return v;
}
bool lean_allowed(int32_t ncol, int64_t nnz) {
    const int mode = lean_setting();
    if (mode == 0) return false;
    return mode == 2 || nnz <= (int64_t)rsp::kLeanMaxMeanLen * (int64_t)ncol;
}

constexpr size_t kRowSlicesFlagBytes = 256;   // the slice form's guard flag, behind the general form's carries
std::atomic<int> g_row_slices{-1};   // rsp_set_row_slices: 0 / 1 / 2; -1 = RSP_ROW_SLICES from the environment, else 1
// 0: never the slice-major form of the row-restricted sums; 1: where it is the faster one; 2: wherever it is possible (tests)
int row_slices_setting() {
    int v = g_row_slices.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = [] {
            const char* s = getenv("RSP_ROW_SLICES");
            const int e = s ? atoi(s) : 1;
            return e <= 0 ? 0 : (e >= 2 ? 2 : 1);
        }();
        v = env;
    }
    return v;
}

std::atomic<int> g_row_segments{-1};   // rsp_set_row_segments: 0 / 1 / 2; -1 = RSP_ROW_SEGMENTS from the environment, else 1
// 0: a handle's row sums never take the segments form; 1: where it is the faster one; 2: wherever it is possible (tests)
int row_segments_setting() {
    int v = g_row_segments.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = [] {
            const char* s = getenv("RSP_ROW_SEGMENTS");
            const int e = s ? atoi(s) : 1;
            return e <= 0 ? 0 : (e >= 2 ? 2 : 1);
        }();
        v = env;
    }
    return v;
}

// columns form of a plan (every column long): RSP_COLUMNS_FORM=0 keeps such matrices on the general kernels (A/B);
// wavefronts per column: RSP_COLUMNS_WAVES = 4 / 8 / 16, else from the mean column length
std::atomic<int> g_columns_form{-1};   // rsp_set_columns_form: 0 / 1 / 2; -1 = RSP_COLUMNS_FORM from the environment, else 1
// 0: never; 1: where it is the faster form (the thresholds below); 2: on every matrix the kernel can take at all (a
// workgroup per column whatever its length: measurements on both sides of the thresholds, tools/edge_sweep.py)
int columns_form_setting() {
    int v = g_columns_form.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = [] {
            const char* s = getenv("RSP_COLUMNS_FORM");
            const int e = s ? atoi(s) : 1;
            return e <= 0 ? 0 : (e >= 2 ? 2 : 1);
        }();
        v = env;
    }
    return v;
}
bool columns_allowed() { return columns_form_setting() != 0; }
int columns_waves_setting(int32_t ncol, int64_t mean_len) {
    static const int env = env_int("RSP_COLUMNS_WAVES");
    if (env == 4 || env == 8 || env == 16) return env;
    // 4 wavefronts per column wherever there are columns enough to fill the chip with them (1000 columns of 3e3 / 1e4 /
    // 3e4 / 1e5 entries: 4 beats 8 and 16 by 1-6 %); fewer columns get more wavefronts each
    (void)mean_len;
    return ncol >= 512 ? 4 : (ncol >= 256 ? 8 : 16);
}

// Rows of x per chunk of the lean form.  A chunk's columns are handed to the 64 lanes of its wavefront, so the chunk
// should hold about 64 of them and rarely more: the largest of 2 / 3 / 4 / 5 / 6 / 8 / 12 / 16 rows that keeps the MEAN
// number of columns per chunk at or below kLeanTargetColumns (C2, 10 per column: 4 rows = 51 columns, never more than
// 61 in a chunk: 17.2 us; 8 rows 18.1, 3 rows 17.4-18.2, 2 rows 20.7 -- profiles/r03_c2.md).  RSP_LEAN_ROWS overrides.
int lean_rows_setting(int32_t ncol, int64_t nnz) {
    static const int env = env_int("RSP_LEAN_ROWS");
    static const int allowed[] = {2, 3, 4, 5, 6, 8, 12, 16};
    for (int a : allowed)
        if (env == a) return a;
    const double mean_len = ncol > 0 ? (double)nnz / (double)ncol : 1.0;
    const double want = rsp::kLeanTargetColumns * mean_len / rsp::kRowElems;
    int rows = allowed[0];
    for (int a : allowed)
        if ((double)a <= want) rows = a;
    return rows;
}

int chunk_rows_setting() {
    const int o = g_chunk_rows_override.load(std::memory_order_relaxed);
    if (o > 0) return o;
    static const int env = env_int("RSP_CHUNK_ROWS");   // thread-safe one-time init
    return env;
}

// Taper of the automatic chunking: the last `permille` thousandths of x are cut into chunks of
// `rows` rows (rsp_set_taper / RSP_TAPER="permille,rows"; -1 = built-in default).
// *chosen: the caller picked the taper (setter or environment), so the "long calls only" rule of the
// automatic choice does not apply.
void taper_setting(int* permille, int* rows, bool* chosen) {
    static const int env_pm = [] {
        const char* s = getenv("RSP_TAPER");
        return s ? atoi(s) : -1;
    }();
    static const int env_rows = [] {
        const char* s = getenv("RSP_TAPER");
        const char* c = s ? strchr(s, ',') : nullptr;
        return c ? atoi(c + 1) : -1;
    }();
    int pm = g_taper_permille.load(std::memory_order_relaxed);
    int r = g_taper_rows.load(std::memory_order_relaxed);
    if (pm < 0) pm = env_pm;
    if (r < 0) r = env_rows;
    *chosen = pm >= 0;
    if (pm < 0) pm = rsp::kTaperPermille;
    if (r <= 0) r = rsp::kTaperRows;
    *permille = pm > 1000 ? 1000 : pm;
    *rows = r;
}

// Chunking policy: enough chunks to keep every CU busy with several waves and
// to let the hardware dispatcher balance the tail, but chunks long enough to
// amortise the per-chunk column search.  Always a whole number of 128-element
// rows so every chunk starts 1 KiB-aligned.  When the call runs for several rounds
// of resident waves, the end of x is cut into shorter chunks (the taper): they are
// dispatched last and fill the chip while the long chunks of the last round finish
// at different times.
rsp::LaunchPlan make_plan(int64_t nnz, bool planned = false) {
    rsp::LaunchPlan plan;
    const int64_t total_rows = (nnz + rsp::kRowElems - 1) / rsp::kRowElems;
    int rows = chunk_rows_setting();
    const bool automatic = rows <= 0;
    if (automatic) {
        const int64_t target_chunks = 256 * 32;   // 256 CUs x 32 waves
        int64_t r = (total_rows + target_chunks - 1) / target_chunks;
        if (r < rsp::kMinChunkRows) r = rsp::kMinChunkRows;
        if (r > 256) r = 256;
        rows = (int)r;
        if (total_rows <= rsp::kShortCallRows && rows < rsp::kShortCallChunkRows) rows = rsp::kShortCallChunkRows;
        // a planned chunk has no column search to amortise: shorter chunks (whole groups of 4 rows) put more
        // wavefronts on the call's drain -- C2 20.5 us at 20 rows, 19.2 at 8 or 12, 21.7 at 6 or 10
        // (profiles/r03_c2.md)
        if (planned && total_rows <= rsp::kShortCallRows) rows = rsp::kPlannedShortCallChunkRows;
    }
    plan.short_pipeline = total_rows <= rsp::kShortCallRows;   // (whatever the chunk length: the call is one round of waves)
    // byte counts and offsets inside one chunk are 32-bit in the kernel (buffer descriptor size,
    // soffset): a chunk never exceeds 1 GiB of x, whatever the knob says
    if (rows > rsp::kMaxChunkRows) rows = rsp::kMaxChunkRows;
    int variant = g_variant.load(std::memory_order_relaxed);
    if (variant < 0) {
        static const int env = env_int("RSP_VARIANT");
        variant = env;
    }
    plan.variant = variant;
    plan.chunk_elems = rows * rsp::kRowElems;
    plan.tail_elems = plan.chunk_elems;
    int64_t nbody = nnz > 0 ? (total_rows + rows - 1) / rows : 0, ntail = 0;
    int pm, trows;
    bool chosen;
    taper_setting(&pm, &trows, &chosen);
    if (automatic && pm > 0 && trows < rows && (chosen || nbody > rsp::kTaperMinChunks)) {
        const int64_t body_rows = (total_rows * (1000 - pm) / 1000) / rows * rows;
        nbody = body_rows / rows;
        ntail = (total_rows - body_rows + trows - 1) / trows;
        plan.tail_elems = trows * rsp::kRowElems;
    }
    plan.nbody = (int32_t)nbody;
    plan.nchunks = (int32_t)(nbody + ntail);
    return plan;
}

// The handle entries run on the handle's device: switch to it for the duration of the call and
// put the calling thread's current device back on every exit path (a torch-hosting process
// would otherwise find its later allocations on another GPU).
// R code forks (parallel::mclapply).  A child of a process that has used the HIP runtime inherits neither a usable GPU context
// nor the runtime's threads; its calls into the runtime may hang.  The host entries therefore remember which process first asked
// for a device, and answer "no device" in a forked child: the Rcpp layer above the ABI then takes its host loop (the
// reference's own), exactly as on a machine without a GPU -- a child never touches the runtime at all.
std::atomic<long> g_hip_pid{0};
bool forked_child() {
    const long me = (long)getpid();
    long seen = g_hip_pid.load(std::memory_order_relaxed);
    if (seen == 0 && g_hip_pid.compare_exchange_strong(seen, me, std::memory_order_relaxed)) return false;
    return seen != me;
}

class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        if (forked_child()) {   // (a handle carried across a fork: fail, do not enter a runtime that may hang)
            err_ = hipErrorNoDevice;
            return;
        }
        if (hipGetDevice(&prev_) != hipSuccess) {
            (void)hipGetLastError();
            prev_ = -1;
        }
        err_ = (prev_ == device) ? hipSuccess : hipSetDevice(device);
        if (prev_ == device) prev_ = -1;   // nothing to restore
    }
    ~DeviceGuard() {
        if (prev_ >= 0) (void)hipSetDevice(prev_);
    }
    hipError_t error() const { return err_; }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
private:
    int prev_ = -1;
    hipError_t err_ = hipSuccess;
};

int check_sizes(int32_t ncol, int64_t nnz) {
    if (ncol < 0) return fail(RSP_ERR_BAD_ARG, "ncol is negative (%d)", ncol);
    if (ncol > INT32_MAX - 65536)   // column cursors run a few windows past the current column in 32 bits
        return fail(RSP_ERR_BAD_ARG, "ncol = %d is above the supported 2^31 - 65537", ncol);
    if (nnz < 0 || nnz > INT32_MAX)
        return fail(RSP_ERR_BAD_ARG,
                    "nnz = %lld is outside [0, 2^31-1] (p[] is 32-bit, RcppSparse.h:30)",
                    (long long)nnz);
    return RSP_OK;
}

// What Matrix::dgCMatrix validity guarantees and the reference loop assumes.
int check_offsets_host(const int32_t* p, int32_t ncol, int64_t nnz) {
    if (p[0] != 0) return fail(RSP_ERR_BAD_ARG, "p[0] = %d, expected 0", p[0]);
    int bad = 0;
    for (int32_t c = 0; c < ncol; ++c) bad |= (p[c + 1] < p[c]);
    if (bad) return fail(RSP_ERR_BAD_ARG, "p[] is not non-decreasing");
    if (p[ncol] != nnz)
        return fail(RSP_ERR_BAD_ARG, "p[ncol] = %d but nnz = %lld", p[ncol], (long long)nnz);
    return RSP_OK;
}

int require_device(int device) {
    if (forked_child())
        return fail(RSP_ERR_NO_DEVICE, "this process was forked from one that had already used the GPU (pid %ld): the HIP runtime does "
                                       "not survive a fork -- no device here", g_hip_pid.load(std::memory_order_relaxed));
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(RSP_ERR_NO_DEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n)
        return fail(RSP_ERR_BAD_ARG, "device %d out of range [0, %d)", device, n);
    return RSP_OK;
}

// ---- the folded fix-up's ticket words (colsums_kernels.hip FOLD): one per (device, launching stream) ----
// A short plain call runs its fix-up inside the main launch; the workgroups count themselves through a device word that
// must be zero when the kernel starts and is zero again when it ends.  Calls on ONE stream run one after the other, so a
// word per stream is enough and two calls can never meet in one; streams beyond kFoldStreams (and capturing streams: a
// graph could be replayed beside another call of its stream's word) take the two-launch form.
// MEASURED AND LEFT OFF (profiles/DEAD_ENDS.md, round 6): C2 through the folded form 37.0 us against 22.5 us in two launches --
// the last workgroup walks ~3900 records with device-scope loads, sixteen dependent passes of ~1.4 us, which costs more
// than the gap between two dependent kernels (4.9 us) ever did.  The form stays selectable for the A/B (tools/ab_c2_fold.sh)
// and is held to the two-launch bits by tests/test_gpu_parity.py::test_folded_fixup_gives_the_two_launch_bits.
std::atomic<int> g_fold_fixup{-1};   // "fold_fixup": 0 (default) never, 1 short plain calls; -1 = RSP_FOLD_FIXUP from the environment
constexpr int kFoldStreams = 64;
struct FoldSlot {
    int device;
    hipStream_t stream;
    uint32_t* word;
};
std::mutex g_fold_mu;
FoldSlot g_fold_slots[kFoldStreams];
int g_fold_used = 0;

int fold_fixup_setting() {
    int v = g_fold_fixup.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = [] {
            const char* s = getenv("RSP_FOLD_FIXUP");
            return (s && s[0]) ? (atoi(s) != 0 ? 1 : 0) : 0;
        }();
        v = env;
    }
    return v;
}

uint32_t* fold_ticket_for(hipStream_t stream) {   // nullptr: this call takes the two-launch form
    if (fold_fixup_setting() == 0) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_fold_mu);
    for (int k = 0; k < g_fold_used; ++k)
        if (g_fold_slots[k].device == device && g_fold_slots[k].stream == stream) return g_fold_slots[k].word;
    if (g_fold_used >= kFoldStreams) return nullptr;
    int on_device = 0;   // (the words live in a __device__ array of each device's copy of the code object)
    for (int k = 0; k < g_fold_used; ++k) on_device += g_fold_slots[k].device == device;
    uint32_t* word = nullptr;
    if (rsp::fold_ticket_address(on_device, &word) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    g_fold_slots[g_fold_used++] = FoldSlot{device, stream, word};
    return word;
}

int enqueue(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz, double* d_out,
            void* d_ws, size_t ws_bytes, double divisor, bool means, hipStream_t stream,
            int op = rsp::kOpSum, const int32_t* d_i = nullptr, const uint32_t* d_bitmap = nullptr,
            int32_t bitmap_words = 0, const int32_t* d_run_if = nullptr) {
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (ncol == 0) return RSP_OK;
    if (!d_p || !d_out || (nnz > 0 && !d_x))
        return fail(RSP_ERR_BAD_ARG, "null device pointer");
    if (((uintptr_t)d_x & 15) != 0)
        return fail(RSP_ERR_BAD_ARG, "d_x must be 16-byte aligned");
    const rsp::LaunchPlan plan = make_plan(nnz);
    const size_t need = rsp::workspace_bytes_for(plan.nchunks);
    if (nnz > 0 && (!d_ws || ws_bytes < need))
        return fail(RSP_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, need);
    uint32_t* fold = nullptr;
    if (plan.short_pipeline && op == rsp::kOpSum && plan.variant == 0 && nnz > 0) fold = fold_ticket_for(stream);
    HIP_TRY(rsp::launch_column_sums(d_x, d_p, ncol, (int32_t)nnz, d_out, plan, d_ws, divisor, means,
                                    stream, op, d_i, d_bitmap, bitmap_words, nullptr, d_run_if, fold));
    return RSP_OK;
}

}  // namespace

// ---- inspector-executor plan of a column-sum call (callers that can show p[] to the host once) ----
struct rsp_colsums_plan {
    int device;
    int32_t ncol;
    int64_t nnz;
    rsp::LaunchPlan lp;      // the chunking the records were made for
    bool snapped;            // every chunk starts at most one group before a column start: the planned kernel applies
    int32_t max_skip;        // largest distance from a chunk's grid start to its first column start
    int2* d_rec;             // nchunks + 1 records {first column, first owned element} on the device
    double inspect_ms;       // host time of the inspection (searches + upload of the records)
    // lean form (every column short): the chunks' headers {first column, columns} and their 16-bit column starts
    bool lean;
    int32_t lean_chunks, lean_stride_dwords, lean_max_columns, lean_rows;
    int2* d_lean_hdr;
    uint32_t* d_lean_offs;
    // columns form (every column long and of similar length): no records at all, one workgroup per column
    bool columns;
    int32_t columns_waves, columns_min, columns_max;
    // device-made plan (rsp_column_sums_plan_create_device): the inspection runs as kernels on the caller's stream and
    // its statistics land in a page-locked host record; until the host has SEEN them (`known`) calls take the general
    // kernels, afterwards the form they select.  d_rec / d_lean_hdr then point into d_mem.
    bool device_built, known;
    void* d_mem;
    size_t d_mem_bytes;      // capacity of d_mem (a recycled allocation may be larger than dl.bytes)
    rsp::DeviceInspectLayout dl;
    rsp::PlanStats* h_stats;
    hipEvent_t ev_begin, ev_end;
};

struct rsp_csc {
    rsp_colsums_plan* plan;   // made at upload (p[] is on the host then); nullptr: not applicable
    int device;
    int32_t nrow, ncol;
    int64_t nnz;
    double* d_x;
    int32_t* d_i;
    int32_t* d_p;
    double* d_out;
    void* d_ws;
    size_t ws_bytes;
    hipStream_t stream;
    // row-wise path: row-major values + offsets, built on the first rowSums and kept
    void* d_row_persist;
    double* d_row_out;
    rsp::RowSumsLayout row_layout;
    rsp::RowSegmentsLayout seg_layout;   // row_segments: the table of (block, column) pieces instead of a regrouped copy
    bool row_segments;
    bool rows_checked;       // the upload has looked at i[]: rows_unsorted says whether some column's rows do not ascend
    int32_t rows_unsorted;
    bool row_ready;
    bool plan_bypass;        // rsp_csc_set_planned(h, 0): the general kernels also where the upload's plan applies (A/B)
    bool borrowed;           // x / i / p belong to the caller (rsp::csc_wrap_device): never freed here
};

extern "C" {

const char* rsp_version(void) { return "rcppsparse_hip 0.1.0 gfx950"; }

const char* rsp_last_error(void) { return g_err; }

int rsp_device_count(int* count) {
    if (!count) return fail(RSP_ERR_BAD_ARG, "count is null");
    if (forked_child()) {   // (a child of a process that used the runtime: see forked_child)
        *count = 0;
        return RSP_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return RSP_OK;
}

int rsp_plan_describe(int64_t nnz, int32_t* plan4) {
    if (!plan4) return fail(RSP_ERR_BAD_ARG, "plan4 is null");
    if (int rc = check_sizes(0, nnz)) return rc;
    const rsp::LaunchPlan plan = make_plan(nnz);
    plan4[0] = plan.chunk_elems;
    plan4[1] = plan.nbody;
    plan4[2] = plan.tail_elems;
    plan4[3] = plan.nchunks;
    return RSP_OK;
}

// ---- measurement / test knobs: ONE entry instead of a setter per knob ----------------
// Process-wide atomics; plans and handles sample them when they are MADE.  Not part of what a production caller
// needs: tools/ and the tests use them for A/B runs and to force a form on both sides of its threshold.
namespace {
int clamp012(int v) { return v <= 0 ? 0 : (v >= 2 ? 2 : 1); }
int auto_plan_setting();
int auto_counter(int which);
std::atomic<int> g_auto_plan{-1};   // "auto_plan": 0 / 1; -1 = RSP_AUTO_PLAN from the environment, else 1
std::atomic<int> g_auto_min_nnz{1 << 20};   // "auto_min_nnz": calls below it never plan for themselves (tests lower it)
}  // namespace

int rsp_debug_set(const char* key, int value) {
    if (!key) return fail(RSP_ERR_BAD_ARG, "key is null");
    const std::string k(key);
    if (k == "chunk_rows") {            // 128-element rows of x per wavefront (0 = automatic; clamped to 1 GiB of x per chunk)
        if (value < 0) return fail(RSP_ERR_BAD_ARG, "chunk_rows is negative");
        g_chunk_rows_override.store(value, std::memory_order_relaxed);
    } else if (k == "taper_permille") { // the last value / 1000 of x in shorter chunks (-1 = built-in default, 0 = no taper)
        if (value > 1000) return fail(RSP_ERR_BAD_ARG, "taper_permille is above 1000");
        g_taper_permille.store(value < 0 ? -1 : value, std::memory_order_relaxed);
    } else if (k == "taper_rows") {     // ... of this many rows (<= 0: built-in default)
        g_taper_rows.store(value <= 0 ? -1 : value, std::memory_order_relaxed);
    } else if (k == "experiment") {     // alternative builds of the main kernel (0 = production)
        if (value < 0) return fail(RSP_ERR_BAD_ARG, "experiment is negative");
        g_variant.store(value, std::memory_order_relaxed);
    } else if (k == "lean") {           // plans: 0 never lean, 1 where it is the faster form, 2 wherever it applies
        g_lean.store(clamp012(value), std::memory_order_relaxed);
    } else if (k == "columns_form") {   // plans: 0 never, 1 where faster, 2 on every matrix the kernel can take
        g_columns_form.store(clamp012(value), std::memory_order_relaxed);
    } else if (k == "row_segments") {   // handles' row sums: 0 never the segments form, 1 where faster, 2 wherever possible
        g_row_segments.store(clamp012(value), std::memory_order_relaxed);
    } else if (k == "row_slices") {     // row-restricted sums: 0 never the slice-major form, 1 where faster, 2 wherever possible
        g_row_slices.store(clamp012(value), std::memory_order_relaxed);
    } else if (k == "auto_plan") {      // the plan-free device entries plan for themselves (1, default) or never (0)
        g_auto_plan.store(value != 0 ? 1 : 0, std::memory_order_relaxed);
    } else if (k == "fold_fixup") {     // short plain calls run their fix-up inside the main launch (1) or as a second launch (0, default: faster)
        g_fold_fixup.store(value != 0 ? 1 : 0, std::memory_order_relaxed);
    } else if (k == "auto_min_nnz") {   // ... for matrices of at least this many entries (default 2^20; the parity suite lowers it to 1)
        g_auto_min_nnz.store(value < 1 ? 1 : value, std::memory_order_relaxed);
    } else {
        return fail(RSP_ERR_BAD_ARG, "unknown knob '%s'", key);
    }
    return RSP_OK;
}

int rsp_debug_get(const char* key, int* value) {
    if (!key || !value) return fail(RSP_ERR_BAD_ARG, "key or value is null");
    const std::string k(key);
    int pm, rows;
    bool chosen;
    if (k == "chunk_rows") *value = chunk_rows_setting();
    else if (k == "taper_permille") { taper_setting(&pm, &rows, &chosen); *value = pm; }
    else if (k == "taper_rows") { taper_setting(&pm, &rows, &chosen); *value = rows; }
    else if (k == "experiment") { const int v = g_variant.load(std::memory_order_relaxed); *value = v < 0 ? env_int("RSP_VARIANT") : v; }
    else if (k == "lean") *value = lean_setting();
    else if (k == "columns_form") *value = columns_form_setting();
    else if (k == "row_segments") *value = row_segments_setting();
    else if (k == "row_slices") *value = row_slices_setting();
    else if (k == "auto_plan") *value = auto_plan_setting();
    else if (k == "auto_min_nnz") *value = g_auto_min_nnz.load(std::memory_order_relaxed);
    else if (k == "fold_fixup") *value = fold_fixup_setting();
    else if (k == "auto_plans_made") *value = auto_counter(0);       // read-only: plans the plan-free entries have made ...
    else if (k == "auto_plans_freed") *value = auto_counter(1);      // ... freed again ...
    else if (k == "auto_plans_retired") *value = auto_counter(2);    // ... retired images waiting for their events right now ...
    else if (k == "auto_plans_recycled") *value = auto_counter(3);   // ... and plans that took a recycled allocation instead of a new one
    else return fail(RSP_ERR_BAD_ARG, "unknown knob '%s'", key);
    return RSP_OK;
}

#ifdef RSP_STAMPS
int rsp_debug_read_stamps(unsigned long long* host, int n) {   // diagnostic build only
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(rsp::read_stamps(host, (size_t)n));
    return RSP_OK;
}
#endif

size_t rsp_column_sums_workspace_bytes(int32_t ncol, int64_t nnz) {
    (void)ncol;
    if (nnz <= 0 || nnz > INT32_MAX) return 256;
    // sized for the chunking in force now; changing rsp_set_tuning afterwards to
    // smaller chunks makes the launch fail with RSP_ERR_WORKSPACE, never overrun
    return rsp::workspace_bytes_for(make_plan(nnz).nchunks);
}

static int auto_enqueue(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz, double* d_out, void* d_ws,
                        size_t ws_bytes, double divisor, bool means, hipStream_t stream);

}  // extern "C"
namespace rsp {
bool process_was_forked_after_gpu_use() { return forked_child(); }   // (multigpu.cpp's host entries ask too)
// the general kernels, no planning: for the library's own one-shot paths (multigpu.cpp: rsp_column_sums_host_multi sees new
// offsets at a fresh address in every call -- nothing to remember)
int column_sums_general(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz, double* d_out, void* d_ws,
                        size_t ws_bytes, hipStream_t stream) {
    return enqueue(d_x, d_p, ncol, nnz, d_out, d_ws, ws_bytes, 1.0, false, stream);
}
}  // namespace rsp
extern "C" {

int rsp_column_sums_device(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz,
                           double* d_sums, void* d_workspace, size_t workspace_bytes, void* stream) {
    return auto_enqueue(d_x, d_p, ncol, nnz, d_sums, d_workspace, workspace_bytes, 1.0, false, (hipStream_t)stream);
}

int rsp_column_means_device(const double* d_x, const int32_t* d_p, int32_t nrow, int32_t ncol,
                            int64_t nnz, double* d_means, void* d_workspace, size_t workspace_bytes,
                            void* stream) {
    return auto_enqueue(d_x, d_p, ncol, nnz, d_means, d_workspace, workspace_bytes, (double)nrow, true,
                        (hipStream_t)stream);
}

// A plan's image goes to the device over a non-blocking stream of its own: a copy on the legacy null stream would
// wait for -- and hold up -- every blocking stream of the process (a host program's, torch's) for the upload.
static hipError_t upload_plan_image(void* dst, const void* src, size_t bytes) {
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return e != hipSuccess ? e : e2;
}

// ---- inspector-executor: plan once (p[] seen by the host), then one launch per call -------------------
static int plan_make(const int32_t* p_host, int32_t ncol, int64_t nnz, int device, rsp_colsums_plan** out) {
    *out = nullptr;
    if (ncol <= 0 || nnz <= 0) return RSP_OK;   // nothing to plan: the general entry handles these shapes
    const auto t0 = std::chrono::steady_clock::now();
    rsp_colsums_plan* pl = new (std::nothrow) rsp_colsums_plan();
    if (!pl) return fail(RSP_ERR_ALLOC, "out of host memory");
    pl->device = device;
    pl->ncol = ncol;
    pl->nnz = nnz;
    pl->lp = make_plan(nnz, true);
    pl->d_rec = nullptr;
    pl->lean = false;
    pl->d_lean_hdr = nullptr;
    pl->d_lean_offs = nullptr;
    pl->columns = false;
    pl->columns_waves = pl->columns_min = pl->columns_max = 0;
    try {
        if (lean_allowed(ncol, nnz)) {
            std::vector<uint32_t> image;
            pl->lean_rows = lean_rows_setting(ncol, nnz);
            const rsp::inspect::LeanLimits lim{rsp::kRowElems, rsp::kLeanMaxColumn, rsp::kLeanMaxColumns};
            if (rsp::inspect::inspect_lean(p_host, ncol, nnz, pl->lean_rows, lim, &image, &pl->lean_chunks,
                                           &pl->lean_stride_dwords, &pl->lean_max_columns)) {
                // headers and offsets in ONE device allocation and ONE copy
                hipError_t e = hipMalloc((void**)&pl->d_lean_hdr, image.size() * 4);
                if (e == hipSuccess) e = upload_plan_image(pl->d_lean_hdr, image.data(), image.size() * 4);
                if (e != hipSuccess) {
                    rsp_column_sums_plan_destroy(pl);
                    return fail(RSP_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e));
                }
                static_assert(sizeof(int2) == sizeof(rsp::inspect::Rec), "a record is two 32-bit numbers");
                pl->d_lean_offs = (uint32_t*)(pl->d_lean_hdr + pl->lean_chunks);
                pl->lean = true;
                pl->snapped = true;     // (the lean form is a planned, one-launch form too)
                pl->max_skip = 0;
            }
        }
        if (!pl->lean) {
            std::vector<rsp::inspect::Rec> rec;
            const rsp::inspect::Grid grid{pl->lp.chunk_elems, pl->lp.nbody, pl->lp.tail_elems, pl->lp.nchunks};
            rsp::inspect::inspect_offsets(p_host, ncol, nnz, grid, &rec, &pl->max_skip);
            pl->snapped = pl->max_skip <= rsp::kGroupElems;
            const bool forced_columns = columns_form_setting() == 2;
            if ((!pl->snapped && columns_allowed()) || forced_columns) {
                // every column long and of similar length: one workgroup per column, nothing to upload
                int32_t mn = INT32_MAX, mx = 0;
                for (int32_t c = 0; c < ncol; ++c) {
                    const int32_t len = p_host[c + 1] - p_host[c];
                    mn = len < mn ? len : mn;
                    mx = len > mx ? len : mx;
                }
                const int64_t mean = nnz / ncol;
                // columns of at least 2048 entries: 4 (8, 16) wavefronts each.  Shorter ones, from 512 entries: 2
                // wavefronts each, and only up to 2.5e8 entries -- a C3 shard (1.25e8 entries in columns of ~1000)
                // 159 -> 149 us, C3 itself 1.238 -> 1.260 ms (the general kernel's long streams win there).
                const bool similar = mx <= rsp::kColumnsMaxLen && (int64_t)mx <= rsp::kColumnsMaxOverMean * mean;
                const bool long_columns = mn >= rsp::kColumnsMinLen;
                const bool mid_columns = mn >= rsp::kColumnsMinLenTwoWaves && nnz <= rsp::kColumnsTwoWavesMaxNnz;
                // (fewer than 128 columns: only while a column is short enough for ONE workgroup to stream it in a few microseconds)
                const bool enough = ncol >= rsp::kColumnsMinColumns || (int64_t)mx <= rsp::kColumnsFewMaxLen + nnz / 192;
                if ((similar && enough && (long_columns || mid_columns)) || (forced_columns && mx <= rsp::kColumnsMaxLen)) {
                    pl->columns = true;
                    pl->snapped = true;   // (a planned, one-launch form too)
                    pl->columns_min = mn;
                    pl->columns_max = mx;
                    pl->columns_waves = long_columns ? columns_waves_setting(ncol, mean) : 2;
                }
            }
            if (pl->snapped && !pl->columns) {
                hipError_t e = hipMalloc((void**)&pl->d_rec, rec.size() * sizeof(int2));
                if (e == hipSuccess) e = upload_plan_image(pl->d_rec, rec.data(), rec.size() * sizeof(int2));
                if (e != hipSuccess) {
                    rsp_column_sums_plan_destroy(pl);
                    return fail(RSP_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e));
                }
            }
        }
    } catch (...) {
        rsp_column_sums_plan_destroy(pl);
        return fail(RSP_ERR_ALLOC, "out of host memory while planning");
    }
    pl->inspect_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = pl;
    return RSP_OK;
}

int rsp_column_sums_plan_create(const int32_t* p, int32_t ncol, int64_t nnz, int device,
                                rsp_colsums_plan_t* plan) {
    if (!plan) return fail(RSP_ERR_BAD_ARG, "plan is null");
    *plan = nullptr;
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (!p) return fail(RSP_ERR_BAD_ARG, "p is null");
    if (int rc = check_offsets_host(p, ncol, nnz)) return rc;
    if (int rc = require_device(device)) return rc;
    DeviceGuard on(device);
    HIP_TRY(on.error());
    rsp_colsums_plan* pl = nullptr;
    if (int rc = plan_make(p, ncol, nnz, device, &pl)) return rc;
    if (!pl) {   // an empty matrix: a plan object that simply sends the call to the general entry
        pl = new (std::nothrow) rsp_colsums_plan();
        if (!pl) return fail(RSP_ERR_ALLOC, "out of host memory");
        pl->device = device;
        pl->ncol = ncol;
        pl->nnz = nnz;
        pl->lp = make_plan(nnz);
        pl->snapped = false;
        pl->max_skip = 0;
        pl->d_rec = nullptr;
        pl->inspect_ms = 0.0;
        pl->lean = false;
        pl->d_lean_hdr = nullptr;
        pl->d_lean_offs = nullptr;
    }
    *plan = pl;
    return RSP_OK;
}

// ---- the same plans for offsets that live in HBM: inspected ON THE DEVICE, on the caller's stream ----------------
// Sizes of the plan memory depend on ncol, nnz and the settings only (never on the offsets' values): the records of
// the planned chunk grid, and -- when the mean column length leaves the lean form possible at all -- the lean grid's
// first columns and an image with room for `capacity` columns per chunk (three times the mean + 16, at least 126;
// a matrix with a denser chunk than that keeps the other forms).
static rsp::DeviceInspectLayout device_plan_layout(int32_t ncol, int64_t nnz, const rsp::LaunchPlan& lp) {
    rsp::DeviceInspectLayout L{};
    size_t off = 0;
    auto take = [&off](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    L.part1_off = take(rsp::kInspectMaxBlocksColumns * sizeof(int4));
    L.part2_off = take(rsp::kInspectMaxBlocksChunks * sizeof(int2));
    L.rec_off = take(((size_t)lp.nchunks + 1) * sizeof(int2));
    L.try_lean = lean_allowed(ncol, nnz) && nnz <= (int64_t)ncol * rsp::kLeanMaxColumn;
    if (L.try_lean) {
        L.lean_rows = lean_rows_setting(ncol, nnz);
        const int64_t chunk = (int64_t)L.lean_rows * rsp::kRowElems;
        const int64_t nchunks = (nnz + chunk - 1) / chunk;
        if (nchunks <= 0 || nchunks > INT32_MAX / 4) {
            L.try_lean = false;
        } else {
            L.lean_chunks = (int32_t)nchunks;
            const int64_t mean = ((int64_t)ncol + nchunks - 1) / nchunks;
            int64_t cap = 3 * mean + 16;
            cap = cap < 126 ? 126 : (cap > rsp::kLeanMaxColumns ? rsp::kLeanMaxColumns : cap);
            L.lean_capacity = (int32_t)cap;
            L.lean_capacity_stride = rsp::inspect::lean_stride_dwords(L.lean_capacity);
            L.first_off = take(((size_t)nchunks + 1) * 4);
            L.hdr_off = take((size_t)nchunks * sizeof(int2) + (size_t)nchunks * (size_t)L.lean_capacity_stride * 4);
        }
    }
    L.bytes = off;
    return L;
}

// The statistics have arrived: choose the form exactly as plan_make does from a host copy of p[].
static void plan_finalize(rsp_colsums_plan* pl) {
    const rsp::PlanStats& st = *pl->h_stats;
    pl->known = true;
    pl->snapped = pl->lean = pl->columns = false;
    pl->max_skip = st.max_skip;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pl->ev_begin, pl->ev_end) == hipSuccess) pl->inspect_ms = (double)ms;   // device time of the inspection
    else (void)hipGetLastError();
    if (st.invalid) return;   // not a dgCMatrix's offsets: the general kernels (in bounds for any p[]) stay
    const int32_t min_len = INT32_MAX - st.inv_min_len, max_len = st.max_len;
    if (pl->dl.try_lean && !st.lean_bad && max_len <= rsp::kLeanMaxColumn && st.lean_widest <= rsp::kLeanMaxColumns) {
        pl->lean = pl->snapped = true;
        pl->max_skip = 0;
        pl->lean_rows = pl->dl.lean_rows;
        pl->lean_chunks = pl->dl.lean_chunks;
        pl->lean_max_columns = st.lean_widest;
        pl->lean_stride_dwords = rsp::inspect::lean_stride_dwords(st.lean_widest);
        pl->d_lean_hdr = (int2*)((char*)pl->d_mem + pl->dl.hdr_off);
        pl->d_lean_offs = (uint32_t*)(pl->d_lean_hdr + pl->lean_chunks);
        return;
    }
    pl->snapped = st.max_skip <= rsp::kGroupElems;
    const bool forced_columns = columns_form_setting() == 2;
    if ((!pl->snapped && columns_allowed()) || forced_columns) {
        const int64_t mean = pl->nnz / pl->ncol;
        const bool similar = max_len <= rsp::kColumnsMaxLen && (int64_t)max_len <= rsp::kColumnsMaxOverMean * mean;
        const bool long_columns = min_len >= rsp::kColumnsMinLen;
        const bool mid_columns = min_len >= rsp::kColumnsMinLenTwoWaves && pl->nnz <= rsp::kColumnsTwoWavesMaxNnz;
        const bool enough = pl->ncol >= rsp::kColumnsMinColumns || (int64_t)max_len <= rsp::kColumnsFewMaxLen + pl->nnz / 192;
        if ((similar && enough && (long_columns || mid_columns)) || (forced_columns && max_len <= rsp::kColumnsMaxLen)) {
            pl->columns = pl->snapped = true;
            pl->columns_min = min_len;
            pl->columns_max = max_len;
            pl->columns_waves = long_columns ? columns_waves_setting(pl->ncol, mean) : 2;
        }
    }
    if (pl->snapped && !pl->columns) pl->d_rec = (int2*)((char*)pl->d_mem + pl->dl.rec_off);
}

// Non-blocking look at a device-made plan's statistics (block: wait for them).  Never called on a capturing stream's
// behalf with anything but a query the capture cannot see: while `stream` is being captured the look is skipped.
static int plan_poll(rsp_colsums_plan* pl, hipStream_t stream, bool block) {
    if (!pl->device_built || pl->known) return RSP_OK;
    if (!block) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
        if (cs != hipStreamCaptureStatusNone) return RSP_OK;
        // the last inspection kernel writes `ready` into the page-locked record behind the statistics (a system-scope
        // release): looking costs one load, touches no runtime state (an event query could disturb a global-mode stream
        // capture of another thread) and cannot fail
        if (__atomic_load_n(&pl->h_stats->ready, __ATOMIC_ACQUIRE) != rsp::inspect::kStatsReady) return RSP_OK;
    } else {
        HIP_TRY(hipEventSynchronize(pl->ev_end));
    }
    plan_finalize(pl);
    return RSP_OK;
}

// What a device-made plan holds of the runtime: freeing any of it WAITS FOR THE DEVICE on this runtime (hipFree,
// hipHostFree and hipFreeAsync of a hipMalloc'ed block all drain every stream first: 21 ms behind 21 ms of queued
// work, tools/microbench notes in profiles/r06_README.md), so the plan-free entries never free -- they recycle.
struct PlanResources {
    int device = -1;
    void* d_mem = nullptr;
    size_t d_mem_bytes = 0;
    rsp::PlanStats* h_stats = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
};

static int plan_create_device_impl(const int32_t* d_p, int32_t ncol, int64_t nnz, hipStream_t s, PlanResources* reuse,
                                   rsp_colsums_plan_t* plan) {
    if (!plan) return fail(RSP_ERR_BAD_ARG, "plan is null");
    *plan = nullptr;
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (!d_p) return fail(RSP_ERR_BAD_ARG, "d_p is null");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    rsp_colsums_plan* pl = new (std::nothrow) rsp_colsums_plan();
    if (!pl) return fail(RSP_ERR_ALLOC, "out of host memory");
    pl->device = device;
    pl->ncol = ncol;
    pl->nnz = nnz;
    if (ncol <= 0 || nnz <= 0) {   // nothing to plan: the general entry handles these shapes
        pl->lp = make_plan(nnz);
        *plan = pl;
        return RSP_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    pl->lp = make_plan(nnz, true);
    pl->device_built = true;
    pl->dl = device_plan_layout(ncol, nnz, pl->lp);
    hipError_t e = hipSuccess;
    if (reuse && reuse->device == device && reuse->d_mem && reuse->d_mem_bytes >= pl->dl.bytes && reuse->h_stats &&
        reuse->ev_begin && reuse->ev_end) {
        pl->d_mem = reuse->d_mem;
        pl->d_mem_bytes = reuse->d_mem_bytes;
        pl->h_stats = reuse->h_stats;
        pl->ev_begin = reuse->ev_begin;
        pl->ev_end = reuse->ev_end;
        *reuse = PlanResources();   // (taken)
    } else {
        e = hipMalloc(&pl->d_mem, pl->dl.bytes);
        if (e == hipSuccess) pl->d_mem_bytes = pl->dl.bytes;
        if (e == hipSuccess) e = hipHostMalloc((void**)&pl->h_stats, sizeof(rsp::PlanStats), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreate(&pl->ev_begin);
        if (e == hipSuccess) e = hipEventCreate(&pl->ev_end);
    }
    if (e == hipSuccess) pl->h_stats->ready = 0;
    if (e == hipSuccess) e = hipEventRecord(pl->ev_begin, s);
    // (the last kernel writes the statistics straight into the page-locked host record: no copy, no memset)
    if (e == hipSuccess) e = rsp::launch_inspect_device(d_p, ncol, (int32_t)nnz, pl->lp, pl->dl, pl->d_mem, pl->h_stats, s);
    if (e == hipSuccess) e = hipEventRecord(pl->ev_end, s);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (pl->ev_end) (void)hipStreamSynchronize(s);   // (whatever was enqueued has to be done before its memory goes)
        rsp_column_sums_plan_destroy(pl);
        return fail(RSP_ERR_HIP, "device-side inspection could not be enqueued: %s", hipGetErrorString(e));
    }
    // (until the statistics have been seen: the host time of the enqueue; afterwards the device time of the kernels)
    pl->inspect_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *plan = pl;
    return RSP_OK;
}

int rsp_column_sums_plan_create_device(const int32_t* d_p, int32_t ncol, int64_t nnz, void* stream,
                                       rsp_colsums_plan_t* plan) {
    return plan_create_device_impl(d_p, ncol, nnz, (hipStream_t)stream, nullptr, plan);
}

// takes a device-made plan apart WITHOUT freeing anything of the runtime's: what it held comes back for the next plan
static PlanResources plan_strip(rsp_colsums_plan* pl) {
    PlanResources r;
    if (pl->device_built) {
        r.device = pl->device;
        r.d_mem = pl->d_mem;
        r.d_mem_bytes = pl->d_mem_bytes;
        r.h_stats = pl->h_stats;
        r.ev_begin = pl->ev_begin;
        r.ev_end = pl->ev_end;
        pl->d_mem = nullptr;
        pl->h_stats = nullptr;
        pl->ev_begin = pl->ev_end = nullptr;
    }
    delete pl;
    return r;
}

static void plan_resources_free(PlanResources& r) {   // WAITS for the device (rsp_release_cached only, and a pool that overflows)
    if (r.device < 0) return;
    DeviceGuard on(r.device);
    if (r.d_mem) (void)hipFree(r.d_mem);
    if (r.h_stats) (void)hipHostFree(r.h_stats);
    if (r.ev_begin) (void)hipEventDestroy(r.ev_begin);
    if (r.ev_end) (void)hipEventDestroy(r.ev_end);
    r = PlanResources();
}

int rsp_column_sums_plan_ready(rsp_colsums_plan_t plan) {
    if (!plan) return -1;
    if (plan->device_built && !plan->known) (void)plan_poll(plan, nullptr, false);
    return (!plan->device_built || plan->known) ? 1 : 0;
}

int rsp_column_sums_plan_wait(rsp_colsums_plan_t plan) {
    if (!plan) return fail(RSP_ERR_BAD_ARG, "null plan");
    return plan_poll(plan, nullptr, true);
}

int rsp_debug_plan_image(rsp_colsums_plan_t plan, int what, void* host, size_t capacity, size_t* bytes) {
    if (!plan || !bytes) return fail(RSP_ERR_BAD_ARG, "null plan or size");
    if (int rc = plan_poll(plan, nullptr, true)) return rc;
    const void* src = nullptr;
    size_t n = 0;
    if (what == 0 && plan->snapped && !plan->lean && !plan->columns && plan->d_rec) {
        src = plan->d_rec;
        n = ((size_t)plan->lp.nchunks + 1) * sizeof(int2);
    } else if (what == 1 && plan->lean && plan->d_lean_hdr) {
        src = plan->d_lean_hdr;
        n = (size_t)plan->lean_chunks * sizeof(int2) + (size_t)plan->lean_chunks * (size_t)plan->lean_stride_dwords * 4;
    }
    *bytes = n;
    if (n == 0 || !host) return RSP_OK;   // (this plan has no such image / the caller only asked for the size)
    if (capacity < n) return fail(RSP_ERR_BAD_ARG, "buffer too small: %zu < %zu bytes", capacity, n);
    DeviceGuard on(plan->device);
    HIP_TRY(on.error());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, src, n, hipMemcpyDeviceToHost));
    return RSP_OK;
}

int rsp_column_sums_plan_info(rsp_colsums_plan_t plan, int32_t* info4, double* inspect_ms) {
    if (!plan || !info4) return fail(RSP_ERR_BAD_ARG, "null plan or output");
    if (int rc = plan_poll(plan, nullptr, true)) return rc;   // (a device-made plan: wait for its inspection)
    info4[0] = plan->columns ? 3 : (plan->lean ? 2 : (plan->snapped ? 1 : 0));
    info4[1] = plan->columns ? plan->ncol : (plan->lean ? plan->lean_chunks : plan->lp.nchunks);
    info4[2] = plan->columns ? plan->columns_waves * 64 : (plan->lean ? plan->lean_rows * rsp::kRowElems : plan->lp.chunk_elems);
    info4[3] = plan->columns ? plan->columns_max : (plan->lean ? plan->lean_max_columns : plan->max_skip);
    if (inspect_ms) *inspect_ms = plan->inspect_ms;
    return RSP_OK;
}

int rsp_column_sums_plan_destroy(rsp_colsums_plan_t plan) {
    if (!plan) return RSP_OK;
    if (forked_child()) return RSP_OK;   // (the parent's, see rsp_csc_free)
    if (plan->device_built) {   // (records and images live inside d_mem)
        DeviceGuard on(plan->device);
        if (plan->ev_end) (void)hipEventSynchronize(plan->ev_end);   // the inspection may still be writing
        if (plan->d_mem) (void)hipFree(plan->d_mem);
        if (plan->h_stats) (void)hipHostFree(plan->h_stats);
        if (plan->ev_begin) (void)hipEventDestroy(plan->ev_begin);
        if (plan->ev_end) (void)hipEventDestroy(plan->ev_end);
    } else if (plan->d_rec || plan->d_lean_hdr) {
        DeviceGuard on(plan->device);
        if (plan->d_rec) (void)hipFree(plan->d_rec);
        if (plan->d_lean_hdr) (void)hipFree(plan->d_lean_hdr);   // (the offsets live behind the headers in the same allocation)
    }
    delete plan;
    return RSP_OK;
}

// ---- the plan-free device entries plan for themselves --------------------------------------------------------
// rsp_column_sums_device / rsp_column_means_device are what a caller with device pointers uses when it knows
// nothing about plans (BASELINE config 2 through them: 0.46-0.51 of the HBM roofline on the general kernels, 0.70
// through a plan).  The library therefore remembers the offsets it has been shown.
//   * The FIRST call on (device, d_p, ncol, nnz) runs the general kernels and only notes the key (a table entry: no
//     allocation, nothing enqueued) -- a caller that shows fresh offsets in every call never pays for a plan.
//   * The SECOND call runs the general kernels too and enqueues the device-side inspection of d_p BEHIND them on the
//     caller's stream (23 us for 1e6 columns; its image and page-locked record are allocated here, outside the table's lock).
//   * Once the host has seen the statistics (the kernels write a `ready` word into the page-locked record: one load per
//     call, no event query, never a wait) calls with that key take the form they select -- lean (every column short: one
//     launch, the reference's bits) or columns (every column long: one launch); matrices with neither keep the general
//     kernels and cost nothing further.  rsp_column_sums_device_settle does all of that at once and WAITS: from its
//     return on every call with the key takes the same form -- bit-identical results run to run.
// Nobody has promised that d_p still holds the offsets that were inspected.  The kernels of this path do not rest
// on it: the lean kernel compares every column's image offsets with the p[] of THIS call and sums a column that
// differs straight from x; the columns kernel reads p[] itself.  Both clamp to [0, nnz] and raise the plan's OWN
// page-locked `stale` word; a call that finds it raised answers with the general kernels, RETIRES the plan and
// inspects again.  So: never a wrong sum, whatever the caller does with d_p.
// A retired plan's image may still be read by launches in flight.  Every plan remembers the streams it was launched
// on; retiring it records an event on the retiring call's stream at once, and the other streams get theirs at their
// next call through these entries; when every such event has completed the image is freed (looked at only while
// something is retired).  Nothing here ever waits for the device, and HBM use is bounded: at most kAutoMaxRetiredPerKey
// retired images per key (a key that has them all outstanding stays on the general kernels until one is freed) and
// kAutoMaxRetired per process.  After kAutoMaxStrikes stale rounds without kAutoForgive clean planned calls in
// between the key stays on the general kernels for good.
// The table's mutex is held for the bookkeeping only: allocations, the inspection's enqueue and the column-sum
// launches themselves run outside it (a plan in use is pinned by a counter).
// A call on a CAPTURING stream records the general kernels and touches none of this: a graph outlives the call, the
// images belong to the library.
namespace {
constexpr int kAutoMaxEntries = 16;
constexpr int kAutoMaxStrikes = 4;
constexpr int kAutoForgive = 32;         // clean planned calls that wipe a key's strikes (a caller rewriting p[] every ~70 calls stays planned)
constexpr int kAutoMaxStreams = 6;         // streams one plan is launched on; a further stream's calls take the general kernels
constexpr int kAutoMaxRetiredPerKey = 2;
constexpr int kAutoMaxRetired = 24;
constexpr uint64_t kAutoIdleTicks = 64;    // a PLANNED entry makes room for a new key only after this many calls without a use
// (calls below 2^20 entries -- g_auto_min_nnz, "auto_min_nnz" -- are launch-bound either way: two launches against one)

// One inspection's result with everything whose lifetime is tied to it.
struct AutoPlan {
    rsp_colsums_plan* plan = nullptr;
    int device = -1;
    int32_t* h_stale = nullptr;            // this plan's own word (page-locked pool): raised by a kernel that found p[] changed
    hipStream_t streams[kAutoMaxStreams];  // streams it has been launched on ...
    int nstreams = 0;
    int pins = 0;                          // launches being issued right now (outside the lock)
    // retired:
    bool retired = false;
    bool awaiting[kAutoMaxStreams];        // ... of which these still owe an event
    hipEvent_t fences[kAutoMaxStreams + 1];
    int nfences = 0;
    struct AutoEntry* owner = nullptr;     // its key's entry while that exists (counts the key's retired images)
};

struct AutoEntry {
    int device = -1;
    const int32_t* d_p = nullptr;
    int32_t ncol = 0;
    int64_t nnz = 0;
    int sightings = 0;
    AutoPlan* cur = nullptr;               // nullptr: none (yet, or given up)
    bool planning = false;                 // a thread is allocating / enqueueing the inspection outside the lock
    bool dead = false;                     // stays on the general kernels for good
    bool want_plan = false;                // a stale round could not inspect again (retired images outstanding): try later
    int nretired = 0;                      // this key's retired images not freed yet
    uint64_t last_use = 0;
    int strikes = 0;
    int clean = 0;                         // planned calls since the last stale round: kAutoForgive of them wipe the strikes
    int last_form = 0;                     // form of the most recent call with this key (rsp_column_sums_device_form)
};
std::mutex g_auto_mu;
std::vector<AutoEntry*> g_auto;            // (pointers: an entry's address survives the table's growth while a thread plans for it)
std::vector<AutoPlan*> g_auto_retired;
uint64_t g_auto_tick = 0;
std::atomic<int> g_auto_made{0}, g_auto_freed{0};   // plans made / freed since the process started (rsp_debug_get: soak runs)
std::atomic<int> g_auto_recycled{0};                // ... of the made ones, how many took a recycled allocation
// what freed plans held of the runtime, waiting for the next plan that fits (same device, at least as many bytes): a
// re-inspection of the same key always fits.  Bounded: kAutoPoolMax sets; a further one replaces the oldest, which IS freed
// there and then (the one place where a call of these entries can wait for the device; it takes more than kAutoPoolMax
// dead images of sizes nobody asks for again)
constexpr size_t kAutoPoolMax = 8;
constexpr size_t kAutoPoolMaxBytes = (size_t)1 << 30;   // ... and at most 1 GiB of HBM in all (a larger image is kept alone)
std::vector<PlanResources> g_plan_pool;

// Another thread of the process may be capturing a stream in GLOBAL mode, which an allocation, an event query or a page-lock
// on ANY thread invalidates.  The few such calls these entries make -- once per key for its plan, and while
// something is retired -- run with THIS thread's capture mode relaxed: they touch nothing a capture could record.
struct RelaxedCaptureMode {
    hipStreamCaptureMode prev = hipStreamCaptureModeRelaxed;
    bool ok = false;
    RelaxedCaptureMode() {
        ok = hipThreadExchangeStreamCaptureMode(&prev) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    ~RelaxedCaptureMode() {
        if (ok && hipThreadExchangeStreamCaptureMode(&prev) != hipSuccess) (void)hipGetLastError();
    }
    RelaxedCaptureMode(const RelaxedCaptureMode&) = delete;
    RelaxedCaptureMode& operator=(const RelaxedCaptureMode&) = delete;
};

// page-locked stale words: 4 bytes each, handed out from whole pages that are never unmapped before
// rsp_release_cached (a word goes back to the free list only when its plan is freed, i.e. when no launch can write it)
std::vector<int32_t*> g_stale_pages, g_stale_free;
int32_t* stale_word_take() {   // caller holds g_auto_mu
    if (g_stale_free.empty()) {
        RelaxedCaptureMode relaxed;
        int32_t* page = nullptr;
        if (hipHostMalloc((void**)&page, 4096, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            if (hipHostMalloc((void**)&page, 4096, hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
        }
        try {
            g_stale_pages.push_back(page);
            for (int k = 0; k < 1024; k += 16) g_stale_free.push_back(page + k);   // one word per 64-byte line
        } catch (...) {
            return nullptr;
        }
    }
    int32_t* w = g_stale_free.back();
    g_stale_free.pop_back();
    *(volatile int32_t*)w = 0;
    return w;
}

int auto_plan_setting() {
    int v = g_auto_plan.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = [] {
            const char* s = getenv("RSP_AUTO_PLAN");
            return (s && s[0]) ? (atoi(s) != 0 ? 1 : 0) : 1;
        }();
        v = env;
    }
    return v;
}

// a plan nothing can read any more (caller holds g_auto_mu): its allocations go to the pool, NOTHING is freed (freeing
// device or page-locked memory waits for every stream of the device on this runtime)
void auto_plan_free(AutoPlan* ap, bool really_free = false) {
    DeviceGuard on(ap->device);
    if (ap->plan) {
        PlanResources r = plan_strip(ap->plan);
        ap->plan = nullptr;
        if (really_free) {
            plan_resources_free(r);
        } else if (r.device >= 0) {
            try {
                size_t held = 0;
                for (const PlanResources& q : g_plan_pool) held += q.d_mem_bytes;
                while (!g_plan_pool.empty() && (g_plan_pool.size() >= kAutoPoolMax || held + r.d_mem_bytes > kAutoPoolMaxBytes)) {
                    held -= g_plan_pool.front().d_mem_bytes;
                    plan_resources_free(g_plan_pool.front());   // (the oldest: really freed, and that waits for the device)
                    g_plan_pool.erase(g_plan_pool.begin());
                }
                g_plan_pool.push_back(r);
            } catch (...) {
                plan_resources_free(r);
            }
        }
    }
    for (int k = 0; k < ap->nfences; ++k) (void)hipEventDestroy(ap->fences[k]);
    if (ap->h_stale) {
        try {
            g_stale_free.push_back(ap->h_stale);
        } catch (...) {   // (the word is simply not handed out again)
        }
    }
    if (ap->owner) --ap->owner->nretired;
    g_auto_freed.fetch_add(1, std::memory_order_relaxed);
    delete ap;
}

// a retired plan owes an event on `stream` if it was launched there (caller holds g_auto_mu, the plan's device is current)
void auto_fence(AutoPlan* ap, hipStream_t stream) {
    for (int k = 0; k < ap->nstreams; ++k) {
        if (ap->streams[k] != stream || !ap->awaiting[k]) continue;
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, stream) == hipSuccess) {
            ap->fences[ap->nfences++] = ev;
            ap->awaiting[k] = false;
        } else {
            (void)hipGetLastError();
            if (ev) (void)hipEventDestroy(ev);   // (it keeps owing: the image stays until rsp_release_cached)
        }
    }
}

void auto_retire(AutoPlan* ap, AutoEntry* owner, hipStream_t stream) {   // caller holds g_auto_mu; ap's device is current
    RelaxedCaptureMode relaxed;
    ap->retired = true;
    ap->owner = owner;
    if (owner) ++owner->nretired;
    for (int k = 0; k < ap->nstreams; ++k) ap->awaiting[k] = true;
    auto_fence(ap, stream);
    try {
        g_auto_retired.push_back(ap);
    } catch (...) {   // (no room to remember it: it is leaked rather than freed under a launch)
    }
}

// every call: streams that owe a retired plan an event pay now; plans nobody can read any more are freed.  Only looks at
// events while something is retired (rare), never waits.
void auto_collect(int device, hipStream_t stream) {   // caller holds g_auto_mu; `device` is current
    RelaxedCaptureMode relaxed;
    for (size_t k = 0; k < g_auto_retired.size();) {
        AutoPlan* ap = g_auto_retired[k];
        bool done = ap->pins == 0;
        if (ap->device == device) auto_fence(ap, stream);
        for (int j = 0; j < ap->nstreams && done; ++j) done = !ap->awaiting[j];
        if (done && ap->device != device) done = false;   // (its events belong to another device: that device's calls look)
        for (int j = 0; j < ap->nfences && done; ++j) {
            const hipError_t q = hipEventQuery(ap->fences[j]);
            if (q != hipSuccess) {
                if (q != hipErrorNotReady) (void)hipGetLastError();
                done = q != hipErrorNotReady;   // (an event that cannot be asked any more holds nothing up)
            }
        }
        if (done) {
            auto_plan_free(ap);
            g_auto_retired[k] = g_auto_retired.back();
            g_auto_retired.pop_back();
        } else {
            ++k;
        }
    }
}

int auto_counter(int which) {
    if (which == 0) return g_auto_made.load(std::memory_order_relaxed);
    if (which == 1) return g_auto_freed.load(std::memory_order_relaxed);
    if (which == 3) return g_auto_recycled.load(std::memory_order_relaxed);
    std::lock_guard<std::mutex> lock(g_auto_mu);
    return (int)g_auto_retired.size();
}

void auto_release_all() {   // rsp_release_cached: the one place that waits for the devices
    std::lock_guard<std::mutex> lock(g_auto_mu);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        ndev = 0;
    }
    {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        for (int d = 0; d < ndev; ++d)
            if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();   // launches that read an image may be in flight on any stream
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    for (AutoEntry* e : g_auto) {
        if (e->cur) {
            e->cur->owner = nullptr;
            auto_plan_free(e->cur, true);
        }
    }
    for (AutoPlan* ap : g_auto_retired) {
        ap->owner = nullptr;
        auto_plan_free(ap, true);
    }
    for (PlanResources& r : g_plan_pool) plan_resources_free(r);
    g_plan_pool.clear();
    for (AutoEntry* e : g_auto) delete e;
    g_auto.clear();
    g_auto_retired.clear();
    g_stale_free.clear();
    for (int32_t* page : g_stale_pages) (void)hipHostFree(page);
    g_stale_pages.clear();
}
}  // namespace

static int planned_enqueue(rsp_colsums_plan_t plan, const double* d_x, const int32_t* d_p, double* d_out,
                           void* d_ws, size_t ws_bytes, double divisor, bool means, hipStream_t stream) {
    if (plan->device_built && !plan->known) {
        // a device-made plan whose inspection the host has not seen yet: look (no waiting); still unknown -> the
        // general kernels answer this call (right for any matrix), the planned form takes over once it is known
        if (int rc = plan_poll(plan, stream, false)) return rc;
        if (!plan->known) return enqueue(d_x, d_p, plan->ncol, plan->nnz, d_out, d_ws, ws_bytes, divisor, means, stream);
    }
    if (plan->lean || plan->snapped) {   // the plan's records live in the HBM of the device it was made on
        int cur = -1;
        HIP_TRY(hipGetDevice(&cur));
        if (cur != plan->device)
            return fail(RSP_ERR_BAD_ARG, "this plan was made on device %d, the calling thread's current device is %d",
                        plan->device, cur);
    }
    if (plan->lean) {   // every column short: rows, header and 16-bit offsets of a chunk requested at once
        if (!d_out || !d_x) return fail(RSP_ERR_BAD_ARG, "null device pointer");
        if (((uintptr_t)d_x & 15) != 0) return fail(RSP_ERR_BAD_ARG, "d_x must be 16-byte aligned");
        HIP_TRY(rsp::launch_column_sums_lean(d_x, (int32_t)plan->nnz, plan->d_lean_hdr, plan->d_lean_offs,
                                             plan->lean_stride_dwords, plan->lean_chunks, plan->lean_rows, d_out, divisor,
                                             means, stream));
        return RSP_OK;
    }
    if (plan->columns) {   // every column long: one workgroup per column (no records; p[] is read by the kernel)
        if (!d_p || !d_out || !d_x) return fail(RSP_ERR_BAD_ARG, "null device pointer");
        HIP_TRY(rsp::launch_column_sums_columns(d_x, d_p, plan->ncol, plan->columns_waves, d_out, divisor, means, stream));
        return RSP_OK;
    }
    if (!plan->snapped)   // a column longer than a group crosses a chunk edge somewhere: the general kernels
        return enqueue(d_x, d_p, plan->ncol, plan->nnz, d_out, d_ws, ws_bytes, divisor, means, stream);
    if (!d_p || !d_out || !d_x) return fail(RSP_ERR_BAD_ARG, "null device pointer");
    if (((uintptr_t)d_x & 15) != 0) return fail(RSP_ERR_BAD_ARG, "d_x must be 16-byte aligned");
    HIP_TRY(rsp::launch_column_sums(d_x, d_p, plan->ncol, (int32_t)plan->nnz, d_out, plan->lp, nullptr, divisor,
                                    means, stream, rsp::kOpSum, nullptr, nullptr, 0, plan->d_rec));
    return RSP_OK;
}

int rsp_column_sums_planned_device(rsp_colsums_plan_t plan, const double* d_x, const int32_t* d_p,
                                   int32_t ncol, int64_t nnz, int32_t nrow_for_means, double* d_sums,
                                   void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!plan || nrow_for_means < 0) return fail(RSP_ERR_BAD_ARG, "null plan or negative nrow_for_means");
    // a plan belongs to the matrix it was made from: the lean form never reads d_p and trusts the plan's sizes,
    // so a matrix of another shape must not get as far as a launch
    if (ncol != plan->ncol || nnz != plan->nnz)
        return fail(RSP_ERR_BAD_ARG, "this plan was made for ncol = %d, nnz = %lld; the call passes ncol = %d, nnz = %lld",
                    plan->ncol, (long long)plan->nnz, ncol, (long long)nnz);
    return planned_enqueue(plan, d_x, d_p, d_sums, d_workspace, workspace_bytes,
                           nrow_for_means > 0 ? (double)nrow_for_means : 1.0, nrow_for_means > 0,
                           (hipStream_t)stream);
}

// Makes the plan of a key: allocates, enqueues the inspection behind what is on `stream` (outside the table's lock).
// nullptr: no memory for it (the key then stays on the general kernels; not an error of the call).
static AutoPlan* auto_make_plan(int device, const int32_t* d_p, int32_t ncol, int64_t nnz, hipStream_t stream, int32_t* h_stale) {
    AutoPlan* ap = new (std::nothrow) AutoPlan();
    if (!ap) return nullptr;
    ap->device = device;
    ap->h_stale = h_stale;
    // a recycled allocation that fits (the smallest such), taken out of the pool under the lock
    PlanResources reuse;
    {
        const size_t need = device_plan_layout(ncol, nnz, make_plan(nnz, true)).bytes;
        std::lock_guard<std::mutex> lock(g_auto_mu);
        size_t best = g_plan_pool.size();
        for (size_t k = 0; k < g_plan_pool.size(); ++k)
            if (g_plan_pool[k].device == device && g_plan_pool[k].d_mem_bytes >= need &&
                (best == g_plan_pool.size() || g_plan_pool[k].d_mem_bytes < g_plan_pool[best].d_mem_bytes))
                best = k;
        if (best < g_plan_pool.size()) {
            reuse = g_plan_pool[best];
            g_plan_pool.erase(g_plan_pool.begin() + (long)best);
        }
    }
    const bool recycled = reuse.device >= 0;
    RelaxedCaptureMode relaxed;   // (a new plan allocates; a recycled one only records events)
    if (plan_create_device_impl(d_p, ncol, nnz, stream, recycled ? &reuse : nullptr, &ap->plan) != RSP_OK) {
        if (reuse.device >= 0) {   // (not taken after all: back into the pool)
            std::lock_guard<std::mutex> lock(g_auto_mu);
            try {
                g_plan_pool.push_back(reuse);
            } catch (...) {
            }
        }
        delete ap;
        return nullptr;
    }
    if (reuse.device >= 0) {   // (offered but not used -- cannot happen after the fit test above; kept safe)
        std::lock_guard<std::mutex> lock(g_auto_mu);
        try {
            g_plan_pool.push_back(reuse);
        } catch (...) {
        }
    }
    g_auto_made.fetch_add(1, std::memory_order_relaxed);
    if (recycled) g_auto_recycled.fetch_add(1, std::memory_order_relaxed);
    return ap;
}

static AutoEntry* auto_find(int device, const int32_t* d_p, int32_t ncol, int64_t nnz) {   // caller holds g_auto_mu
    for (AutoEntry* c : g_auto)
        if (c->device == device && c->d_p == d_p && c->ncol == ncol && c->nnz == nnz) return c;
    return nullptr;
}

// A place in the table for a new key, or nullptr (everything in use).  Entries that never got a plan go first (they hold
// nothing); a planned one only after kAutoIdleTicks calls without a use -- its plan is retired, never waited for.
static AutoEntry* auto_new_entry(int device, const int32_t* d_p, int32_t ncol, int64_t nnz, hipStream_t stream) {   // caller holds g_auto_mu
    if ((int)g_auto.size() >= kAutoMaxEntries) {
        size_t victim = g_auto.size();
        for (size_t k = 0; k < g_auto.size(); ++k) {
            AutoEntry* c = g_auto[k];
            if (c->planning || c->nretired > 0) continue;
            const bool bare = c->cur == nullptr;
            if (!bare && (g_auto_tick - c->last_use < kAutoIdleTicks || (int)g_auto_retired.size() >= kAutoMaxRetired)) continue;
            if (victim == g_auto.size()) { victim = k; continue; }
            AutoEntry* v = g_auto[victim];
            const bool vbare = v->cur == nullptr;
            if ((bare && !vbare) || (bare == vbare && c->last_use < v->last_use)) victim = k;
        }
        if (victim == g_auto.size()) return nullptr;
        AutoEntry* v = g_auto[victim];
        if (v->cur) {
            DeviceGuard on(v->cur->device);
            auto_retire(v->cur, nullptr, v->cur->device == device ? stream : nullptr);
        }
        delete v;
        g_auto[victim] = g_auto.back();
        g_auto.pop_back();
    }
    AutoEntry* ne = new (std::nothrow) AutoEntry();
    if (!ne) return nullptr;
    ne->device = device;
    ne->d_p = d_p;
    ne->ncol = ncol;
    ne->nnz = nnz;
    try {
        g_auto.push_back(ne);
    } catch (...) {
        delete ne;
        return nullptr;
    }
    return ne;
}

static int auto_enqueue(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz, double* d_out, void* d_ws,
                        size_t ws_bytes, double divisor, bool means, hipStream_t stream) {
    if (auto_plan_setting() == 0 || nnz < (int64_t)g_auto_min_nnz.load(std::memory_order_relaxed) || ncol <= 0 || nnz > INT32_MAX || !d_p || !d_x || !d_out ||
        ((uintptr_t)d_x & 15) != 0)
        return enqueue(d_x, d_p, ncol, nnz, d_out, d_ws, ws_bytes, divisor, means, stream);
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
    // A capture records the general kernels, whatever is known about these offsets: a graph outlives this call, and the
    // images of the entry's own plans belong to the library (a retirement would leave the graph a dangling pointer).
    // A caller that wants the planned form in a graph makes the plan itself and owns its lifetime.
    if (cs != hipStreamCaptureStatusNone) return enqueue(d_x, d_p, ncol, nnz, d_out, d_ws, ws_bytes, divisor, means, stream);

    // ---- bookkeeping under the lock: which form, and what else this call has to do ----
    enum { kGeneral, kLean, kColumns } form = kGeneral;
    AutoPlan* use = nullptr;       // pinned: launched below, outside the lock
    AutoEntry* plan_for = nullptr; // this call makes the key's plan (behind its own general launch)
    int32_t* new_stale = nullptr;
    rsp_colsums_plan snap{};       // what the launch needs of the plan (copied: the entry may move on meanwhile)
    {
        std::lock_guard<std::mutex> lock(g_auto_mu);
        ++g_auto_tick;
        if (!g_auto_retired.empty()) auto_collect(device, stream);
        AutoEntry* e = auto_find(device, d_p, ncol, nnz);
        if (!e) e = auto_new_entry(device, d_p, ncol, nnz, stream);   // first sighting: the key is noted, nothing else
        if (e) {
            e->last_use = g_auto_tick;
            ++e->sightings;
            AutoPlan* ap = e->cur;
            if (ap && *(volatile int32_t*)ap->h_stale != 0) {
                // a kernel found p[] changed under the plan: this call on the general kernels, the plan retired (launches in
                // flight may still read its image), a fresh inspection behind this call
                auto_retire(ap, e, stream);
                e->cur = nullptr;
                ap = nullptr;
                e->last_form = 0;
                e->clean = 0;
                if (++e->strikes >= kAutoMaxStrikes) e->dead = true;
                e->want_plan = !e->dead;
            }
            if (ap && !ap->plan->known) (void)plan_poll(ap->plan, stream, false);
            if (ap && ap->plan->known && (ap->plan->lean || ap->plan->columns)) {
                int slot = -1;
                for (int k = 0; k < ap->nstreams; ++k)
                    if (ap->streams[k] == stream) slot = k;
                if (slot < 0 && ap->nstreams < kAutoMaxStreams) {
                    slot = ap->nstreams++;
                    ap->streams[slot] = stream;
                }
                if (slot >= 0) {   // (a seventh stream on one plan: the general kernels answer its calls)
                    use = ap;
                    ++ap->pins;
                    snap = *ap->plan;
                    form = ap->plan->lean ? kLean : kColumns;
                    if (++e->clean >= kAutoForgive) e->strikes = 0;
                }
            }
            e->last_form = form == kLean ? 2 : (form == kColumns ? 3 : 0);
            const bool wants = !e->cur && !e->dead && !e->planning && (e->sightings >= 2 || e->want_plan);
            if (wants && e->nretired < kAutoMaxRetiredPerKey && (new_stale = stale_word_take()) != nullptr) {
                e->planning = true;
                plan_for = e;
            }
        }
    }

    // ---- the launches, outside the lock ----
    int rc = RSP_OK;
    hipError_t le = hipSuccess;
    if (form == kLean) {
        le = rsp::launch_column_sums_lean(d_x, (int32_t)nnz, snap.d_lean_hdr, snap.d_lean_offs, snap.lean_stride_dwords,
                                          snap.lean_chunks, snap.lean_rows, d_out, divisor, means, stream, d_p, ncol, use->h_stale);
    } else if (form == kColumns) {
        // lengths the choice of the form rested on, with room: a column outside [min / 4, 4 max] says the matrix has changed
        const int32_t lo = snap.columns_min / 4, hi = snap.columns_max > rsp::kColumnsMaxLen / 4 ? rsp::kColumnsMaxLen : 4 * snap.columns_max;
        le = rsp::launch_column_sums_columns(d_x, d_p, ncol, snap.columns_waves, d_out, divisor, means, stream, (int32_t)nnz, lo, hi,
                                             use->h_stale);
    } else {
        rc = enqueue(d_x, d_p, ncol, nnz, d_out, d_ws, ws_bytes, divisor, means, stream);
    }
    if (le != hipSuccess) rc = fail(RSP_ERR_HIP, "planned column-sum launch failed: %s", hipGetErrorString(le));
    AutoPlan* made = nullptr;
    if (plan_for && rc == RSP_OK) made = auto_make_plan(device, d_p, ncol, nnz, stream, new_stale);

    // ---- and the lock once more, only if something has to be put back ----
    if (use || plan_for) {
        std::lock_guard<std::mutex> lock(g_auto_mu);
        if (use) {
            --use->pins;
            if (use->retired) {   // retired while this launch was being issued: its event comes now
                RelaxedCaptureMode relaxed;
                auto_fence(use, stream);
            }
        }
        if (plan_for) {
            plan_for->planning = false;
            plan_for->want_plan = false;
            if (made) plan_for->cur = made;
            else {
                try {
                    g_stale_free.push_back(new_stale);
                } catch (...) {
                }
                if (rc == RSP_OK) plan_for->dead = true;   // no memory for an image: the general kernels for good
            }
        }
    }
    return rc;
}

int rsp_column_sums_device_settle(const int32_t* d_p, int32_t ncol, int64_t nnz, void* stream) {
    if (!d_p || ncol <= 0 || nnz < 0) return -1;
    if (auto_plan_setting() == 0 || nnz < (int64_t)g_auto_min_nnz.load(std::memory_order_relaxed) || nnz > INT32_MAX) return 0;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    AutoEntry* plan_for = nullptr;
    int32_t* new_stale = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_auto_mu);
        ++g_auto_tick;
        AutoEntry* e = auto_find(device, d_p, ncol, nnz);
        if (!e) e = auto_new_entry(device, d_p, ncol, nnz, s);
        if (!e) return 0;   // the table is full of keys in use: this one stays on the general kernels (until one goes idle)
        e->last_use = g_auto_tick;
        if (e->dead) return 0;
        if (!e->cur && !e->planning && e->nretired < kAutoMaxRetiredPerKey && (new_stale = stale_word_take()) != nullptr) {
            e->planning = true;
            plan_for = e;
        }
    }
    if (plan_for) {
        AutoPlan* made = auto_make_plan(device, d_p, ncol, nnz, s, new_stale);
        std::lock_guard<std::mutex> lock(g_auto_mu);
        plan_for->planning = false;
        plan_for->want_plan = false;
        plan_for->sightings = plan_for->sightings < 2 ? 2 : plan_for->sightings;
        if (made) plan_for->cur = made;
        else {
            try {
                g_stale_free.push_back(new_stale);
            } catch (...) {
            }
            plan_for->dead = true;
        }
    }
    return rsp_column_sums_device_form(d_p, ncol, nnz, 1);
}

int rsp_column_sums_device_form(const int32_t* d_p, int32_t ncol, int64_t nnz, int wait) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (int attempt = 0;; ++attempt) {
        rsp_colsums_plan* pl = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_auto_mu);
            AutoEntry* c = auto_find(device, d_p, ncol, nnz);
            if (!c) return -1;
            if (c->dead) return 0;                                   // given up on: the general kernels
            if (!c->cur) {
                if (!(c->planning && wait && attempt < 2000)) return -1;   // (another thread is making the plan right now)
            } else {
                pl = c->cur->plan;
                if (!pl->known) (void)plan_poll(pl, nullptr, false);
                if (pl->known) return pl->lean ? 2 : (pl->columns ? 3 : 0);   // (a snapped plan is not taken here: general kernels)
                if (!wait) return -1;
                ++c->cur->pins;   // (the plan cannot go away while this thread waits for its inspection)
            }
        }
        if (!pl) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            continue;
        }
        hipError_t e;
        {
            RelaxedCaptureMode relaxed;   // (waiting for an event is one of the calls a global-mode capture elsewhere forbids)
            e = hipEventSynchronize(pl->ev_end);
        }
        std::lock_guard<std::mutex> lock(g_auto_mu);
        AutoEntry* c = auto_find(device, d_p, ncol, nnz);
        AutoPlan* holder = nullptr;
        if (c && c->cur && c->cur->plan == pl) holder = c->cur;
        for (AutoPlan* ap : g_auto_retired)
            if (ap->plan == pl) holder = ap;
        if (holder) --holder->pins;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        if (!holder || holder->retired) continue;   // retired meanwhile: look again
        if (!pl->known) plan_finalize(pl);
        return pl->lean ? 2 : (pl->columns ? 3 : 0);
    }
}

int rsp_column_reduce_device(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz, int op,
                             double* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    int kop;
    switch (op) {
        case RSP_OP_SUM: kop = rsp::kOpSum; break;
        case RSP_OP_SUM_SQUARES: kop = rsp::kOpSumSquares; break;
        case RSP_OP_SUM_ABS: kop = rsp::kOpSumAbs; break;
        case RSP_OP_MAX: kop = rsp::kOpMax; break;
        case RSP_OP_MIN: kop = rsp::kOpMin; break;
        case RSP_OP_COUNT: kop = rsp::kOpCount; break;
        default: return fail(RSP_ERR_BAD_ARG, "unknown reduction op %d", op);
    }
    return enqueue(d_x, d_p, ncol, nnz, d_out, d_workspace, workspace_bytes, 1.0, false,
                   (hipStream_t)stream, kop);
}

int rsp_column_sums_in_rows_device(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t nrow,
                                   int32_t ncol, int64_t nnz, const uint32_t* d_row_bitmap, int complement,
                                   double* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (nrow < 0) return fail(RSP_ERR_BAD_ARG, "nrow is negative");
    if (nnz > 0 && (!d_i || !d_row_bitmap)) return fail(RSP_ERR_BAD_ARG, "row indices and row bitmap are required");
    if (((uintptr_t)d_i & 7) != 0) return fail(RSP_ERR_BAD_ARG, "d_i must be 8-byte aligned");
    const int32_t words = (int32_t)(((int64_t)nrow + 31) / 32);
    const int op = complement ? rsp::kOpMaskedOut : rsp::kOpMaskedIn;
    // More than 2^20 rows and tens of entries per column and slice: the slice-major form (bitmap probed in LDS),
    // with the general form standing by for matrices its device-side guard turns away.  The guard's flag lives
    // behind the general form's carries: a workspace of rsp_column_sums_in_rows_workspace_bytes has room for it,
    // one sized by rsp_column_sums_workspace_bytes selects the general form alone.
    rsp::RowSlicesPlan sp;
    if (row_slices_setting() != 0 && nnz > 0 && ncol > 0 && nnz <= INT32_MAX && d_x && d_p && d_out && d_workspace &&
        rsp::rowslices_applicable(nrow, ncol, nnz, row_slices_setting() == 2, &sp)) {
        const size_t general = rsp::workspace_bytes_for(make_plan(nnz).nchunks);
        if (workspace_bytes >= general + kRowSlicesFlagBytes) {
            int32_t* d_flag = (int32_t*)((char*)d_workspace + general);
            HIP_TRY(rsp::launch_rowslices_guard(d_p, ncol, sp, d_flag, (hipStream_t)stream));
            HIP_TRY(rsp::launch_column_sums_rowslices(d_x, d_i, d_p, ncol, (int32_t)nnz, d_row_bitmap, words,
                                                      complement != 0, sp, d_out, d_flag, (hipStream_t)stream));
            return enqueue(d_x, d_p, ncol, nnz, d_out, d_workspace, general, 1.0, false, (hipStream_t)stream, op,
                           d_i, d_row_bitmap, words, d_flag);
        }
    }
    return enqueue(d_x, d_p, ncol, nnz, d_out, d_workspace, workspace_bytes, 1.0, false, (hipStream_t)stream, op, d_i,
                   d_row_bitmap, words);
}

size_t rsp_column_sums_in_rows_workspace_bytes(int32_t nrow, int32_t ncol, int64_t nnz) {
    (void)nrow;
    return rsp_column_sums_workspace_bytes(ncol, nnz) + kRowSlicesFlagBytes;
}

int rsp_column_sums_in_rows_form(int32_t nrow, int32_t ncol, int64_t nnz, size_t workspace_bytes) {
    if (nrow < 0 || ncol < 0 || nnz < 0 || nnz > INT32_MAX) return -1;
    const size_t bitmap_bytes = (size_t)(((int64_t)nrow + 31) / 32) * 4;
    if (bitmap_bytes <= rsp::kLdsBitmapMinBytes) return RSP_IN_ROWS_FORM_L1;
    if (bitmap_bytes <= rsp::kLdsBitmapMaxBytes) return RSP_IN_ROWS_FORM_LDS;
    if (row_slices_setting() != 0 && nnz > 0 && ncol > 0 && rsp::rowslices_applicable(nrow, ncol, nnz, row_slices_setting() == 2, nullptr) &&
        workspace_bytes >= rsp::workspace_bytes_for(make_plan(nnz).nchunks) + kRowSlicesFlagBytes)
        return RSP_IN_ROWS_FORM_SLICES;
    return RSP_IN_ROWS_FORM_L2;
}

int rsp_column_sums_device_timed(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz,
                                 double* d_sums, void* d_workspace, size_t workspace_bytes,
                                 void* stream, int reps, float* ms_per_call) {
    if (reps <= 0 || !ms_per_call) return fail(RSP_ERR_BAD_ARG, "reps <= 0 or null result");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = RSP_OK;
    hipError_t he = hipEventRecord(e0, s);
    for (int r = 0; r < reps && rc == RSP_OK && he == hipSuccess; ++r)
        rc = enqueue(d_x, d_p, ncol, nnz, d_sums, d_workspace, workspace_bytes, 1.0, false, s);
    if (he == hipSuccess) he = hipEventRecord(e1, s);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != RSP_OK) return rc;
    if (he != hipSuccess) return fail(RSP_ERR_HIP, "timing failed: %s", hipGetErrorString(he));
    *ms_per_call = ms / (float)reps;
    return RSP_OK;
}

int rsp_debug_exclusive_scan_device(const int32_t* d_in, int32_t* d_out, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_debug_exclusive_scan_device");
    if (n == 0) return RSP_OK;
    const size_t bytes = rsp::exclusive_scan_temp_bytes(n);
    void* temp = nullptr;
    HIP_TRY(hipMalloc(&temp, bytes));
    hipError_t e = rsp::launch_exclusive_scan_i32(d_in, d_out, n, 0, temp, bytes, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(temp);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "exclusive scan failed: %s", hipGetErrorString(e));
    return RSP_OK;
}

int rsp_debug_read_ceiling_device(const double* d_x, int64_t nnz, double* d_sink, void* stream, int reps,
                                  float* ms_per_launch) {
    if (int rc = check_sizes(0, nnz)) return rc;
    if (reps <= 0 || !ms_per_launch) return fail(RSP_ERR_BAD_ARG, "reps <= 0 or null result");
    if (!d_sink || (nnz > 0 && !d_x)) return fail(RSP_ERR_BAD_ARG, "null device pointer");
    if (((uintptr_t)d_x & 15) != 0) return fail(RSP_ERR_BAD_ARG, "d_x must be 16-byte aligned");
    const rsp::LaunchPlan plan = make_plan(nnz);   // the chunk grid rsp_column_sums_device would use
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipError_t he = rsp::launch_read_ceiling(d_x, (int32_t)nnz, plan, d_sink, s);   // one untimed launch first
    if (he == hipSuccess) he = hipEventRecord(e0, s);
    for (int r = 0; r < reps && he == hipSuccess; ++r) he = rsp::launch_read_ceiling(d_x, (int32_t)nnz, plan, d_sink, s);
    if (he == hipSuccess) he = hipEventRecord(e1, s);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(RSP_ERR_HIP, "read ceiling failed: %s", hipGetErrorString(he));
    *ms_per_launch = ms / (float)reps;
    return RSP_OK;
}

int rsp_gen_values_device(double* d_x, int64_t n, uint64_t seed, uint64_t first_idx, int kind,
                          void* stream) {
    if (n < 0 || (n > 0 && !d_x)) return fail(RSP_ERR_BAD_ARG, "bad buffer");
    if (kind != 0 && kind != 1) return fail(RSP_ERR_BAD_ARG, "kind must be 0 or 1");
    HIP_TRY(rsp::launch_gen_values(d_x, n, seed, first_idx, kind, (hipStream_t)stream));
    return RSP_OK;
}

int rsp_gen_row_indices_device(int32_t* d_i, const int32_t* d_p, int32_t nrow, int32_t ncol,
                               uint64_t seed, void* stream) {
    if (ncol < 0 || nrow < 0 || (ncol > 0 && (!d_i || !d_p))) return fail(RSP_ERR_BAD_ARG, "bad buffer");
    HIP_TRY(rsp::launch_gen_row_indices(d_i, d_p, nrow, ncol, seed, (hipStream_t)stream));
    return RSP_OK;
}

// ---- device-resident dgCMatrix -------------------------------------------

int rsp_csc_free(rsp_csc_t h) {
    if (!h) return RSP_OK;
    if (forked_child()) return RSP_OK;   // (a handle carried across a fork is the parent's: nothing to release here, no runtime to enter)
    DeviceGuard on(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_x && !h->borrowed) (void)hipFree(h->d_x);
    if (h->d_i && !h->borrowed) (void)hipFree(h->d_i);
    if (h->d_p && !h->borrowed) (void)hipFree(h->d_p);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->d_ws) (void)hipFree(h->d_ws);
    if (h->d_row_persist) (void)hipFree(h->d_row_persist);
    if (h->d_row_out) (void)hipFree(h->d_row_out);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->plan) rsp_column_sums_plan_destroy(h->plan);
    delete h;
    return RSP_OK;
}

static int csc_upload(const double* x, const int32_t* i, const int32_t* p, int32_t nrow, int32_t ncol,
                      int64_t nnz, int device, bool with_plan, rsp_csc_t* handle) {
    if (!handle) return fail(RSP_ERR_BAD_ARG, "handle is null");
    *handle = nullptr;
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (nrow < 0) return fail(RSP_ERR_BAD_ARG, "nrow is negative");
    if (!p || (nnz > 0 && !x)) return fail(RSP_ERR_BAD_ARG, "x or p is null");
    if (int rc = check_offsets_host(p, ncol, nnz)) return rc;
    if (int rc = require_device(device)) return rc;
    DeviceGuard on(device);
    HIP_TRY(on.error());

    rsp_csc* h = new (std::nothrow) rsp_csc();
    if (!h) return fail(RSP_ERR_ALLOC, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->nrow = nrow;
    h->ncol = ncol;
    h->nnz = nnz;
    h->ws_bytes = rsp_column_sums_workspace_bytes(ncol, nnz);
    // x is padded to a whole 16-byte pair so the device copy never ends mid-load
    const size_t xbytes = ((size_t)nnz * 8 + 15) & ~(size_t)15;
    // p[] is in host memory right now: inspect it once, so that every columnSums on this handle is one launch
    // without column search, carries or fix-up wherever the matrix allows (no column longer than a group across
    // a chunk edge); a plan that does not apply costs nothing later.  The inspection is host work (1.6 ms for C2's
    // 1e6 columns, 0.1-0.3 s for 1e8) and runs on a thread of its own beside the copies below, which keep this
    // thread busy staging pageable memory; its own small upload goes over the null stream.
    // (the one-shot host entry sums once: an inspection of 1e7 columns costs as much as uploading them)
    rsp_colsums_plan* planned = nullptr;
    std::thread inspector;
    bool inspector_started = false;
    // From 65536 columns on the inspection runs ON THE DEVICE behind the copy of p[] (round 5): the host inspector costs
    // ~6 us per 1000 columns and was what an upload of a C2-sized matrix waited for (copies 1.7 ms, inspector 8.0 ms); the
    // device inspector is 23 us for 1e6 columns.  Same images bit for bit; its lean image has room for 3 x the mean
    // number of columns per chunk + 16 (at least 126), so a matrix with a denser chunk takes the snapped / general form here.
    constexpr int32_t kUploadDeviceInspectMinCols = 65536;
    const bool device_inspect = with_plan && ncol >= kUploadDeviceInspectMinCols && nnz > 0;
    if (with_plan && !device_inspect) {
        try {
            inspector = std::thread([&planned, p, ncol, nnz, device] {
                if (hipSetDevice(device) != hipSuccess) return;
                if (plan_make(p, ncol, nnz, device, &planned) != RSP_OK) planned = nullptr;
            });
            inspector_started = true;
        } catch (...) {
            inspector_started = false;   // (no thread to be had: inspect afterwards, on this one)
        }
    }
    static const bool timing = env_int("RSP_UPLOAD_TIMING") != 0;   // (phases of an upload to stderr: tools/measure_upload.py)
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [&t_begin] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_x, xbytes ? xbytes : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_p, ((size_t)ncol + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_out, ncol ? (size_t)ncol * 8 : 8);
    if (e == hipSuccess) e = hipMalloc(&h->d_ws, h->ws_bytes);
    if (e == hipSuccess && i && nnz > 0) e = hipMalloc((void**)&h->d_i, (size_t)nnz * 4);
    const double t_alloc = ms_since();
    if (e == hipSuccess && nnz > 0)
        e = hipMemcpyAsync(h->d_x, x, (size_t)nnz * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(h->d_p, p, ((size_t)ncol + 1) * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && h->d_i)
        e = hipMemcpyAsync(h->d_i, i, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream);
    // row sums in the segments form need every column's rows to ascend: one pass over i[] in HBM, behind the copies
    // (0.4 ms for 5e8 entries, next to an upload of ~100 ms), only for shapes that form could serve
    if (e == hipSuccess && h->d_i && row_segments_setting() != 0 &&
        rsp::row_segments_applicable(nrow, ncol, nnz, row_segments_setting() == 2)) {
        int32_t* d_flag = (int32_t*)h->d_ws;   // (the column sums' workspace is idle until the first call)
        e = rsp::launch_rows_sorted_check(h->d_i, h->d_p, ncol, nnz, d_flag, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&h->rows_unsorted, d_flag, 4, hipMemcpyDeviceToHost, h->stream);
        h->rows_checked = e == hipSuccess;
    }
    if (e == hipSuccess && device_inspect &&
        rsp_column_sums_plan_create_device(h->d_p, ncol, nnz, h->stream, &planned) != RSP_OK)
        planned = nullptr;   // (no memory for the images: the handle simply runs the general kernels)
    const double t_enqueued = ms_since();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);   // host buffers are only borrowed
    if (e == hipSuccess && device_inspect && planned && plan_poll(planned, nullptr, true) != RSP_OK) {
        rsp_column_sums_plan_destroy(planned);
        planned = nullptr;
    }
    // The device-made lean image is sized BEFORE the offsets are seen (3 x the mean number of columns per chunk + 16, at
    // least 126).  A matrix of short columns with one denser chunk does not fit it and would take the snapped / general
    // kernels here -- within tolerance -- although the host inspector gives it the lean form and with it the reference's
    // bits, as it does below 65536 columns.  With every column <= 64 entries `lean_bad` can only mean
    // "more columns in a chunk than the image has room for" (no such column reaches a row past its chunk): then p[], which is
    // in host memory right here, is inspected again by the host inspector, whose image has the room the matrix needs.
    if (e == hipSuccess && device_inspect && planned && planned->known && !planned->lean && planned->dl.try_lean &&
        planned->h_stats && !planned->h_stats->invalid && planned->h_stats->lean_bad &&
        planned->h_stats->max_len <= rsp::kLeanMaxColumn && planned->h_stats->lean_widest <= rsp::kLeanMaxColumns) {
        rsp_colsums_plan* by_host = nullptr;
        if (plan_make(p, ncol, nnz, device, &by_host) == RSP_OK && by_host && by_host->lean) {
            rsp_column_sums_plan_destroy(planned);
            planned = by_host;
        } else if (by_host) {
            rsp_column_sums_plan_destroy(by_host);
        }
    }
    const double t_copied = ms_since();
    if (inspector_started) inspector.join();
    if (timing)
        fprintf(stderr, "[rsp upload] nnz %lld ncol %d: allocations %.3f ms, copies enqueued %.3f, copies done %.3f, inspector joined %.3f\n",
                (long long)nnz, ncol, t_alloc, t_enqueued, t_copied, ms_since());
    if (e != hipSuccess) {
        if (planned) rsp_column_sums_plan_destroy(planned);
        rsp_csc_free(h);
        return fail(RSP_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    }
    if (with_plan && !device_inspect && !inspector_started && plan_make(p, ncol, nnz, device, &planned) != RSP_OK) planned = nullptr;
    h->plan = planned;
    *handle = h;
    return RSP_OK;
}

int rsp_csc_upload(const double* x, const int32_t* i, const int32_t* p, int32_t nrow, int32_t ncol,
                   int64_t nnz, int device, rsp_csc_t* handle) {
    return csc_upload(x, i, p, nrow, ncol, nnz, device, true, handle);
}

// the handle's column sums (means) enqueued on its own stream into `d_out` (the handle's device is current): nothing waits
static int csc_enqueue(rsp_csc_t h, double* d_out, bool means) {
    if (h->plan && h->plan->snapped && !h->plan_bypass)
        return planned_enqueue(h->plan, h->d_x, h->d_p, d_out, h->d_ws, h->ws_bytes, means ? (double)h->nrow : 1.0, means,
                               h->stream);
    return enqueue(h->d_x, h->d_p, h->ncol, h->nnz, d_out, h->d_ws, h->ws_bytes, means ? (double)h->nrow : 1.0, means,
                   h->stream);
}

static int csc_run(rsp_csc_t h, double* host_out, bool means) {
    if (!h || !host_out) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    DeviceGuard on(h->device);
    HIP_TRY(on.error());
    if (h->ncol == 0) return RSP_OK;
    if (int rc = csc_enqueue(h, h->d_out, means)) return rc;
    HIP_TRY(hipMemcpyAsync(host_out, h->d_out, (size_t)h->ncol * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RSP_OK;
}

}  // extern "C"
namespace rsp {
// ---- what multigpu.cpp needs of a resident shard (the struct stays private to this file) ----
int csc_view(rsp_csc_t h, CscView* v) {
    if (!h || !v) return fail(RSP_ERR_BAD_ARG, "null handle");
    v->device = h->device;
    v->nrow = h->nrow;
    v->ncol = h->ncol;
    v->nnz = h->nnz;
    v->d_out = h->d_out;
    v->stream = h->stream;
    return RSP_OK;
}

int csc_enqueue_columns(rsp_csc_t h, bool means, double* d_out) {
    if (!h) return fail(RSP_ERR_BAD_ARG, "null handle");
    if (h->ncol == 0) return RSP_OK;
    return csc_enqueue(h, d_out ? d_out : h->d_out, means);
}

// A shard over device memory the CALLER owns (x, p and optionally i already in the HBM of `device`): nothing is copied,
// nothing of the caller's is ever freed; the handle adds its own stream, output, workspace and a device-made plan
// (inspected on the handle's stream and waited for, so that every later call takes its final form).
int csc_wrap_device(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t nrow, int32_t ncol, int64_t nnz,
                    int device, rsp_csc_t* handle) {
    if (!handle) return fail(RSP_ERR_BAD_ARG, "handle is null");
    *handle = nullptr;
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (nrow < 0) return fail(RSP_ERR_BAD_ARG, "nrow is negative");
    if (!d_p || (nnz > 0 && !d_x)) return fail(RSP_ERR_BAD_ARG, "d_x or d_p is null");
    if (((uintptr_t)d_x & 15) != 0) return fail(RSP_ERR_BAD_ARG, "d_x must be 16-byte aligned");
    if (int rc = require_device(device)) return rc;
    DeviceGuard on(device);
    HIP_TRY(on.error());
    rsp_csc* h = new (std::nothrow) rsp_csc();
    if (!h) return fail(RSP_ERR_ALLOC, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->nrow = nrow;
    h->ncol = ncol;
    h->nnz = nnz;
    h->borrowed = true;
    h->d_x = const_cast<double*>(d_x);
    h->d_i = const_cast<int32_t*>(d_i);
    h->d_p = const_cast<int32_t*>(d_p);
    h->ws_bytes = rsp_column_sums_workspace_bytes(ncol, nnz);
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_out, ncol ? (size_t)ncol * 8 : 8);
    if (e == hipSuccess) e = hipMalloc(&h->d_ws, h->ws_bytes);
    if (e != hipSuccess) {
        rsp_csc_free(h);
        return fail(RSP_ERR_HIP, "wrapping the device shard failed: %s", hipGetErrorString(e));
    }
    rsp_colsums_plan* planned = nullptr;
    if (ncol > 0 && nnz > 0 && rsp_column_sums_plan_create_device(d_p, ncol, nnz, h->stream, &planned) == RSP_OK) {
        if (plan_poll(planned, nullptr, true) != RSP_OK) {
            rsp_column_sums_plan_destroy(planned);
            planned = nullptr;
        }
    }
    h->plan = planned;
    *handle = h;
    return RSP_OK;
}
}  // namespace rsp
extern "C" {

int rsp_csc_column_sums(rsp_csc_t h, double* sums) { return csc_run(h, sums, false); }

int rsp_csc_dims(rsp_csc_t h, int32_t* nrow, int32_t* ncol, int64_t* nnz) {
    if (!h) return fail(RSP_ERR_BAD_ARG, "null handle");
    if (nrow) *nrow = h->nrow;
    if (ncol) *ncol = h->ncol;
    if (nnz) *nnz = h->nnz;
    return RSP_OK;
}

int rsp_csc_column_form(rsp_csc_t h) {
    if (!h) return -1;
    if (!h->plan || !h->plan->snapped || h->plan_bypass) return 0;
    return h->plan->columns ? 3 : (h->plan->lean ? 2 : 1);
}

int rsp_csc_set_planned(rsp_csc_t h, int on) {
    if (!h) return fail(RSP_ERR_BAD_ARG, "null handle");
    h->plan_bypass = on == 0;
    return RSP_OK;
}
int rsp_csc_column_means(rsp_csc_t h, double* means) { return csc_run(h, means, true); }

// ---- row-wise "next" entries (Matrix::rowSums / rowMeans, RcppSparse.h:138-156) ----------

static int row_plan(int32_t nrow, int64_t nnz, bool keep_row_form, rsp::RowSumsLayout* L) {
    if (nrow < 0) return fail(RSP_ERR_BAD_ARG, "nrow is negative");
    if (int rc = check_sizes(0, nnz)) return rc;
    const size_t cs = rsp::workspace_bytes_for(make_plan(nnz).nchunks);
    hipError_t e = rsp::plan_row_sums(nrow, nnz, cs, keep_row_form, L);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RSP_ERR_HIP, "planning the row-wise path failed: %s", hipGetErrorString(e));
    }
    return RSP_OK;
}

size_t rsp_row_sums_workspace_bytes(int32_t nrow, int64_t nnz) {
    rsp::RowSumsLayout L;
    if (row_plan(nrow, nnz, false, &L) != RSP_OK) return 0;
    return L.persistent_bytes + L.scratch_bytes;
}

static int row_enqueue(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz, double* d_out,
                       void* ws, size_t ws_bytes, double divisor, bool means, hipStream_t stream) {
    if (nrow == 0) return RSP_OK;
    if (!d_out || (nnz > 0 && (!d_x || !d_i))) return fail(RSP_ERR_BAD_ARG, "null device pointer");
    rsp::RowSumsLayout L;
    if (int rc = row_plan(nrow, nnz, false, &L)) return rc;
    if (!ws || ws_bytes < L.persistent_bytes + L.scratch_bytes)
        return fail(RSP_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes,
                    L.persistent_bytes + L.scratch_bytes);
    char* persist = (char*)ws;
    char* scratch = persist + L.persistent_bytes;
    HIP_TRY(rsp::launch_row_build(d_x, d_i, nrow, nnz, L, persist, scratch, stream));
    HIP_TRY(rsp::launch_row_reduce(d_x, d_i, nrow, nnz, L, persist, d_out, divisor, means, make_plan(nnz), stream));
    return RSP_OK;
}

int rsp_row_sums_device(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz, double* d_sums,
                        void* d_workspace, size_t workspace_bytes, void* stream) {
    return row_enqueue(d_x, d_i, nrow, nnz, d_sums, d_workspace, workspace_bytes, 1.0, false,
                       (hipStream_t)stream);
}

int rsp_row_means_device(const double* d_x, const int32_t* d_i, int32_t nrow, int32_t ncol, int64_t nnz,
                         double* d_means, void* d_workspace, size_t workspace_bytes, void* stream) {
    return row_enqueue(d_x, d_i, nrow, nnz, d_means, d_workspace, workspace_bytes, (double)ncol, true,
                       (hipStream_t)stream);
}

// builds the handle's row form on first use and enqueues its row sums (means) on the handle's stream into d_out (nullptr: the
// handle's own output); the handle's device must be current; nothing waits once the form exists
static int csc_rows_enqueue(rsp_csc_t h, double* d_out, bool means) {
    if (h->nrow == 0) return RSP_OK;
    if (h->nnz > 0 && !h->d_i)
        return fail(RSP_ERR_BAD_ARG, "this handle was uploaded without i[]: rowSums needs the row indices");
    if (!h->row_ready) {
        // Long columns whose rows ascend (checked here, once, on the device): the segments form -- a table of every
        // column's piece per row block, and the accumulate pass reads the uploaded x / i as they are (12 B/nnz, no
        // regrouped copy).  Otherwise: the row-major form is built once; its scratch is released right after.
        const int mode = row_segments_setting();
        if (mode != 0 && h->d_i && h->d_p && rsp::row_segments_applicable(h->nrow, h->ncol, h->nnz, mode == 2)) {
            hipError_t e = rsp::plan_row_segments(h->nrow, h->ncol, h->nnz, &h->seg_layout);
            int32_t unsorted = h->rows_checked ? h->rows_unsorted : 1;
            if (e == hipSuccess && !h->d_row_persist) e = hipMalloc(&h->d_row_persist, h->seg_layout.bytes);
            if (e == hipSuccess && !h->d_row_out) e = hipMalloc((void**)&h->d_row_out, (size_t)h->nrow * 8);
            if (e == hipSuccess && !h->rows_checked) {   // (the setting was changed after the upload)
                int32_t* d_flag = (int32_t*)((char*)h->d_row_persist + h->seg_layout.flag_off);
                e = rsp::launch_rows_sorted_check(h->d_i, h->d_p, h->ncol, h->nnz, d_flag, h->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(&unsorted, d_flag, 4, hipMemcpyDeviceToHost, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            }
            if (e == hipSuccess && unsorted == 0) {
                e = rsp::launch_row_segments_build(h->d_i, h->d_p, h->ncol, h->nnz, h->seg_layout, h->d_row_persist,
                                                   h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                if (e == hipSuccess) {
                    h->row_segments = true;
                    h->row_ready = true;
                }
            }
            if (!h->row_ready) {   // rows that do not ascend, or no memory for the table: the general forms below
                if (h->d_row_persist) (void)hipFree(h->d_row_persist);
                h->d_row_persist = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    if (!h->row_ready) {
        if (int rc = row_plan(h->nrow, h->nnz, true, &h->row_layout)) return rc;
        void* scratch = nullptr;
        hipError_t e = hipSuccess;
        if (!h->d_row_persist) e = hipMalloc(&h->d_row_persist, h->row_layout.persistent_bytes);
        if (e == hipSuccess && !h->d_row_out) e = hipMalloc((void**)&h->d_row_out, (size_t)h->nrow * 8);
        if (e == hipSuccess) e = hipMalloc(&scratch, h->row_layout.scratch_bytes);
        if (e == hipSuccess)
            e = rsp::launch_row_build(h->d_x, h->d_i, h->nrow, h->nnz, h->row_layout, h->d_row_persist, scratch,
                                      h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (scratch) (void)hipFree(scratch);
        if (e != hipSuccess) {   // leave the handle usable for the column entries and for a retry
            if (h->d_row_persist) (void)hipFree(h->d_row_persist);
            if (h->d_row_out) (void)hipFree(h->d_row_out);
            h->d_row_persist = nullptr;
            h->d_row_out = nullptr;
            (void)hipGetLastError();
            return fail(RSP_ERR_HIP, "building the row-major form failed: %s", hipGetErrorString(e));
        }
        h->row_segments = false;
        h->row_ready = true;
    }
    double* out = d_out ? d_out : h->d_row_out;
    if (h->row_segments)
        HIP_TRY(rsp::launch_row_segments_reduce(h->d_x, h->d_i, h->nrow, h->ncol, h->seg_layout, h->d_row_persist,
                                                out, means ? (double)h->ncol : 1.0, means, h->stream));
    else
        HIP_TRY(rsp::launch_row_reduce(h->d_x, h->d_i, h->nrow, h->nnz, h->row_layout, h->d_row_persist, out,
                                       means ? (double)h->ncol : 1.0, means, make_plan(h->nnz), h->stream));
    return RSP_OK;
}

static int csc_rows(rsp_csc_t h, double* host_out, bool means) {
    if (!h || !host_out) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    DeviceGuard on(h->device);
    HIP_TRY(on.error());
    if (h->nrow == 0) return RSP_OK;
    if (int rc = csc_rows_enqueue(h, nullptr, means)) return rc;
    HIP_TRY(hipMemcpyAsync(host_out, h->d_row_out, (size_t)h->nrow * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RSP_OK;
}

}  // extern "C"
namespace rsp {
// the shard's PARTIAL row sums (its own columns' entries) enqueued on its stream into d_out, nrow doubles in its device's HBM
// (multigpu.cpp: the single-process handle reduces the shards' vectors on the devices); the shard's device must be current
int csc_enqueue_rows(rsp_csc_t h, double* d_out) {
    if (!h || !d_out) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    return csc_rows_enqueue(h, d_out, false);
}
}  // namespace rsp
extern "C" {


int rsp_csc_row_sums(rsp_csc_t h, double* sums) { return csc_rows(h, sums, false); }

int rsp_csc_row_form(rsp_csc_t h) {
    if (!h) return -1;
    if (!h->row_ready) return RSP_ROW_FORM_NONE;
    if (h->row_segments) return RSP_ROW_FORM_SEGMENTS;
    if (h->row_layout.direct) return RSP_ROW_FORM_DIRECT;
    return h->row_layout.mode == 3 ? RSP_ROW_FORM_TWO_LEVEL : RSP_ROW_FORM_PARTITION;
}

int rsp_csc_row_means(rsp_csc_t h, double* means) { return csc_rows(h, means, true); }

// ---- Matrix::crossprod (RcppSparse.h:159-194) -----------------------------------------------

static bool crossprod_exact() {
    int v = g_crossprod_exact.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* env = getenv("RSP_CROSSPROD_EXACT");
        v = (env && env[0] && env[0] != '0') ? 1 : 0;
        g_crossprod_exact.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

int rsp_set_crossprod_exact(int exact) {
    g_crossprod_exact.store(exact ? 1 : 0, std::memory_order_relaxed);
    return RSP_OK;
}

static int xp_plan(int32_t nrow, int32_t ncol, int64_t nnz, rsp::CrossprodLayout* L) {
    if (nrow < 0) return fail(RSP_ERR_BAD_ARG, "nrow is negative");
    if (int rc = check_sizes(ncol, nnz)) return rc;
    hipError_t e = rsp::plan_crossprod(nrow, ncol, nnz, crossprod_exact(), L);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RSP_ERR_HIP, "planning crossprod failed: %s", hipGetErrorString(e));
    }
    return RSP_OK;
}

size_t rsp_crossprod_workspace_bytes(int32_t nrow, int32_t ncol, int64_t nnz) {
    rsp::CrossprodLayout L;
    if (xp_plan(nrow, ncol, nnz, &L) != RSP_OK) return 0;
    return L.total_bytes;
}

int rsp_crossprod_form(int32_t nrow, int32_t ncol, int64_t nnz) {
    rsp::CrossprodLayout L;
    if (xp_plan(nrow, ncol, nnz, &L) != RSP_OK) return -1;
    return L.tall ? RSP_CROSSPROD_FORM_TALL : RSP_CROSSPROD_FORM_EXACT;
}

int rsp_crossprod_device(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t nrow, int32_t ncol,
                         int64_t nnz, double* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (nrow < 0) return fail(RSP_ERR_BAD_ARG, "nrow is negative");
    if (ncol == 0) return RSP_OK;
    if (!d_p || !d_out || (nnz > 0 && (!d_x || !d_i))) return fail(RSP_ERR_BAD_ARG, "null device pointer");
    if (!d_workspace) {   // no scratch: the tile kernel
        HIP_TRY(rsp::launch_crossprod(d_x, d_i, d_p, ncol, d_out, (hipStream_t)stream));
        return RSP_OK;
    }
    rsp::CrossprodLayout L;
    if (int rc = xp_plan(nrow, ncol, nnz, &L)) return rc;
    if (workspace_bytes < L.total_bytes)
        return fail(RSP_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, L.total_bytes);
    HIP_TRY(rsp::launch_crossprod_rows(d_x, d_i, d_p, nrow, ncol, nnz, d_out, L, d_workspace,
                                       (hipStream_t)stream));
    return RSP_OK;
}

int rsp_csc_crossprod(rsp_csc_t h, double* out) {
    if (!h || !out) return fail(RSP_ERR_BAD_ARG, "null handle or output");
    DeviceGuard on(h->device);
    HIP_TRY(on.error());
    if (h->ncol == 0) return RSP_OK;
    if (h->nnz > 0 && !h->d_i)
        return fail(RSP_ERR_BAD_ARG, "this handle was uploaded without i[]: crossprod needs the row indices");
    rsp::CrossprodLayout L;
    if (int rc = xp_plan(h->nrow, h->ncol, h->nnz, &L)) return rc;
    const size_t bytes = (size_t)h->ncol * (size_t)h->ncol * 8;
    double* d_c = nullptr;
    void* d_ws = nullptr;
    hipError_t e = hipMalloc((void**)&d_c, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_ws, L.total_bytes);
    // (this call synchronises anyway: the tall form runs alone, and only if it reports a sum that is not finite do the
    // exact kernels run -- instead of standing by in every call, eight launches that do nothing)
    int32_t not_finite = 0;
    if (e == hipSuccess)
        e = rsp::launch_crossprod_rows(h->d_x, h->d_i, h->d_p, h->nrow, h->ncol, h->nnz, d_c, L, d_ws, h->stream,
                                       L.tall ? rsp::kXpTallOnly : rsp::kXpAll);
    if (e == hipSuccess && L.tall)
        e = hipMemcpyAsync(&not_finite, (char*)d_ws + L.flag_off, 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_c, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && L.tall && not_finite) {
        e = rsp::launch_crossprod_rows(h->d_x, h->d_i, h->d_p, h->nrow, h->ncol, h->nnz, d_c, L, d_ws, h->stream,
                                       rsp::kXpExactOnly);
        if (e == hipSuccess) e = hipMemcpyAsync(out, d_c, bytes, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (d_ws) (void)hipFree(d_ws);
    if (d_c) (void)hipFree(d_c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RSP_ERR_HIP, "crossprod failed: %s", hipGetErrorString(e));
    }
    return RSP_OK;
}

// ---- one-shot host entry: a grow-only arena per device instead of stream + 4 x hipMalloc / hipFree per call ----
// The exported columnSums(A) of the R package lands here once per call (reference src/example.cpp:26-32 is one
// synchronous call).  Creating a stream, four allocations and four frees around every call cost more than the
// transfers and the kernels of any matrix below ~1e7 entries (round 4: C2 one-shot 3.7 ms against 1.9 ms of
// upload + 0.23 ms of resident call), so the library keeps ONE stream and ONE set of buffers per device between
// calls and only ever grows them.  Calls that need more than kOneShotKeepBytes in total (RSP_ONE_SHOT_KEEP_MB,
// default 1 GiB) allocate what they need for the call and give it back: next to a transfer of that size the
// allocations are noise, and an R session should not sit on 8 GB of HBM because it once summed a big matrix.
// rsp_release_cached() frees everything the library keeps; the R package calls it from R_unload_RcppSparse.
// Thread safety: the arena of a device is held under its mutex for the duration of a call -- concurrent one-shot
// calls on ONE device take turns, calls on different devices run side by side.
namespace {
struct OneShotArena {
    std::mutex mu;
    hipStream_t stream = nullptr;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};   // x, p, sums, workspace
    size_t cap[4] = {0, 0, 0, 0};
};
constexpr int kMaxArenaDevices = 64;
OneShotArena g_arena[kMaxArenaDevices];

size_t one_shot_keep_bytes() {
    static const size_t v = [] {
        const char* s = getenv("RSP_ONE_SHOT_KEEP_MB");
        const long long mb = s ? atoll(s) : 1024;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return v;
}

void arena_release(OneShotArena& a) {   // caller holds a.mu and has made the device current
    if (a.stream) (void)hipStreamSynchronize(a.stream);
    for (int k = 0; k < 4; ++k) {
        if (a.buf[k]) (void)hipFree(a.buf[k]);
        a.buf[k] = nullptr;
        a.cap[k] = 0;
    }
    if (a.stream) (void)hipStreamDestroy(a.stream);
    a.stream = nullptr;
}
}  // namespace

int rsp_release_cached(void) {
    if (forked_child()) return RSP_OK;   // (what is cached belongs to the parent process's runtime)
    auto_release_all();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return RSP_OK;   // no runtime, nothing kept
    }
    for (int d = 0; d < n && d < kMaxArenaDevices; ++d) {
        std::lock_guard<std::mutex> lock(g_arena[d].mu);
        if (!g_arena[d].stream && !g_arena[d].buf[0] && !g_arena[d].buf[1] && !g_arena[d].buf[2] && !g_arena[d].buf[3]) continue;
        DeviceGuard on(d);
        if (on.error() != hipSuccess) continue;
        arena_release(g_arena[d]);
    }
    return RSP_OK;
}

int rsp_column_sums_host(const double* x, const int32_t* p, int32_t ncol, int64_t nnz, double* sums,
                         int device) {
    if (!sums && ncol > 0) return fail(RSP_ERR_BAD_ARG, "sums is null");
    if (int rc = check_sizes(ncol, nnz)) return rc;
    if (!p || (nnz > 0 && !x)) return fail(RSP_ERR_BAD_ARG, "x or p is null");
    if (int rc = check_offsets_host(p, ncol, nnz)) return rc;
    if (int rc = require_device(device)) return rc;
    if (ncol == 0) return RSP_OK;
    if (device >= kMaxArenaDevices) return fail(RSP_ERR_BAD_ARG, "device %d is beyond the %d devices the one-shot entry keeps buffers for", device, kMaxArenaDevices);
    DeviceGuard on(device);
    HIP_TRY(on.error());
    // x is padded to a whole 16-byte pair so the device copy never ends mid-load
    const size_t need[4] = {(((size_t)nnz * 8 + 15) & ~(size_t)15) + 16, ((size_t)ncol + 1) * 4, (size_t)ncol * 8,
                            rsp_column_sums_workspace_bytes(ncol, nnz)};
    OneShotArena& a = g_arena[device];
    std::lock_guard<std::mutex> lock(a.mu);
    hipError_t e = hipSuccess;
    if (!a.stream) e = hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking);
    const bool keep = need[0] + need[1] + need[2] + need[3] <= one_shot_keep_bytes();
    void* own[4] = {nullptr, nullptr, nullptr, nullptr};   // buffers of a call too large to keep
    void* buf[4];
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
        if (keep) {
            if (a.cap[k] < need[k]) {   // grow: a quarter more than asked, so a sequence of slightly larger matrices settles
                if (a.buf[k]) (void)hipFree(a.buf[k]);
                a.buf[k] = nullptr;
                a.cap[k] = 0;
                const size_t want = need[k] + need[k] / 4 + 256;
                e = hipMalloc(&a.buf[k], want);
                if (e == hipSuccess) a.cap[k] = want;
            }
            buf[k] = a.buf[k];
        } else {
            e = hipMalloc(&own[k], need[k]);
            buf[k] = own[k];
        }
    }
    int rc = RSP_OK;
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(buf[0], x, (size_t)nnz * 8, hipMemcpyHostToDevice, a.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(buf[1], p, need[1], hipMemcpyHostToDevice, a.stream);
    if (e == hipSuccess)
        rc = enqueue((const double*)buf[0], (const int32_t*)buf[1], ncol, nnz, (double*)buf[2], buf[3], need[3], 1.0, false,
                     a.stream);
    if (e == hipSuccess && rc == RSP_OK) e = hipMemcpyAsync(sums, buf[2], need[2], hipMemcpyDeviceToHost, a.stream);
    const hipError_t es = hipStreamSynchronize(a.stream);   // host buffers are only borrowed: nothing of this call is left in flight
    if (e == hipSuccess) e = es;
    for (int k = 0; k < 4; ++k)
        if (own[k]) (void)hipFree(own[k]);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RSP_ERR_HIP, "one-shot column sums failed: %s", hipGetErrorString(e));
    }
    return rc;
}

}  // extern "C"

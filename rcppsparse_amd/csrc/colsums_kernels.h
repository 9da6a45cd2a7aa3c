// colsums_kernels.h -- shared constants and launcher prototypes (internal).
#ifndef RSP_COLSUMS_KERNELS_H
#define RSP_COLSUMS_KERNELS_H

#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include "inspect.hpp"

namespace rsp {

constexpr int kRowElems = 128;    // one wave instruction: 64 lanes x 16 B = 128 doubles
constexpr int kBatchRows = 8;     // rows (1 KiB loads) kept in flight per wavefront
constexpr int kMinChunkRows = 16; // automatic chunking never goes below this many rows per wave
constexpr int kWavesPerWG = 4;    // independent wavefronts per workgroup
constexpr int kPWin = 256;        // p[] entries staged in LDS per wavefront
constexpr int kHistPad = 132;     // 129 histogram slots, padded to a 16-byte multiple
constexpr int kStampChunks = 65536; // diagnostic build: chunks that record timestamps
constexpr int kLoadAux = 2;       // buffer_load cache policy: 2 = nt (streamed once)
constexpr int kGroupRows = 4;     // rows per dense group (512 elements, 8 per lane)
constexpr int kGroupElems = kGroupRows * kRowElems;
constexpr int kStageSlots = kGroupElems + 32;  // LDS staging of one group (+pad: unguarded look-ahead reads)
constexpr int kDenseMaxEnds = 2048; // more column ends than this in one group: general path
constexpr int kDenseMaxLen = 64;  // longest run of elements one lane may sum alone in a dense group
constexpr int kFewEnds = 3;       // rows with <= this many column ends use the masked-reduce loop
#ifndef RSP_DENSE_MIN_ENDS      // (-DRSP_DENSE_MIN_ENDS=n builds a variant for threshold sweeps, tools/ab.py)
#define RSP_DENSE_MIN_ENDS 4
#endif
constexpr int kDenseMinEnds = RSP_DENSE_MIN_ENDS;  // groups with >= this many column ends use the dense path

// per-element transforms of the generic column reduction (values of rsp_column_reduce_device's op)
constexpr int kOpSum = 0;
constexpr int kOpSumSquares = 1;
constexpr int kOpSumAbs = 2;
constexpr int kOpMaskedIn = 3;    // only entries whose row is in a row set (bitmap)
constexpr int kOpMaskedOut = 4;   // only entries whose row is NOT in the set
constexpr int kOpMax = 5;         // largest stored entry (combine = max, identity -inf)
constexpr int kOpMin = 6;         // smallest stored entry
constexpr int kOpCount = 7;       // number of stored entries (offsets only)

// How a column-sum call is cut into chunks (one wavefront each).  The first `nbody` chunks have
// `chunk_elems` elements; the rest of x (the tapered tail) is cut into shorter chunks of
// `tail_elems` elements, which the dispatcher hands out last: they fill the chip while the long
// chunks of the final round finish at different times.  Without a taper
// tail_elems == chunk_elems.  Chunk starts are multiples of kRowElems (1 KiB of x).
struct LaunchPlan {
    int32_t chunk_elems;   // elements per body chunk, multiple of kRowElems
    int32_t nbody;         // number of body chunks
    int32_t tail_elems;    // elements per tail chunk, multiple of kRowElems
    int32_t nchunks;       // nbody + tail chunks; >= 1 when nnz > 0
    bool short_pipeline;   // 4 rows in flight per wavefront instead of 8 (calls that fit one round of waves)
    int variant;           // 0 = production kernel; >0 = experiment variants (env RSP_VARIANT)
};
// Calls of at most this many 128-element rows (C2: 78 125) are one round of resident wavefronts that all
// start together: half of x is requested in the first microsecond and nothing is consumed before it
// has arrived.  With 4 rows in flight (and 20-row chunks) the first rows are there sooner and the
// column work starts earlier: C2 26.6 -> 24.6 us (profiles/r02_c2.md); longer calls keep 8.
constexpr int kShortCallRows = 131072;
constexpr int kShortCallChunkRows = 20;
constexpr int kPlannedShortCallChunkRows = 8;   // inspector-executor form of such calls: no search to amortise
// lean planned kernel (all columns short): chunk of 2..16 rows (capi.hip lean_rows_setting); a chunk may hold up to kLeanMaxColumns columns,
// no column is longer than kLeanMaxColumn entries or reaches more than one row past its chunk's grid end
constexpr int kLeanTargetColumns = 52;                    // mean number of columns per chunk the row count aims at
constexpr int kLeanMaxColumn = 64;
constexpr int kLeanMaxMeanLen = 60;                       // mean column length up to which the form is selected (rsp_set_lean(2): any)
constexpr int kLeanXcdRun = 16;                          // neighbouring workgroups that go to the same XCD (measured: 4 / 8 / 16 / 32)
constexpr int kLeanMaxColumns = 1278;                     // + 1 closing offset + 1 pad = 1280 16-bit offsets
constexpr int kLeanMaxOffsetDwords = 640;                 // = 10 x 64 lanes
constexpr int kGuessWindow = 1024;   // offsets read around the guessed first column of a chunk (short calls)
constexpr int kTaperPermille = 100;  // default taper: the last 10 % of x ...
constexpr int kTaperRows = 64;       // ... in chunks of 64 rows (when the body's chunks are longer)
constexpr int kTaperMinChunks = 12288; // only calls of more than two rounds of resident waves are tapered (a 1.25e8-nnz shard loses 1 % with it)
constexpr size_t kLdsBitmapMinBytes = 16 * 1024;   // row-restricted sums: smaller bitmaps stay in L1 ...
constexpr size_t kLdsBitmapMaxBytes = 128 * 1024;  // ... bigger ones do not fit beside the kernel's 21.5 KB of LDS
constexpr int kMaxChunkRows = 1 << 20;   // 1 GiB of x per chunk: byte counts and offsets inside a chunk stay < 2^31

// chunk index <-> first element, shared by host code and both kernels
struct ChunkMap {
    int32_t body, nbody, tail;
    __host__ __device__ int64_t start(int32_t w) const {
        return w < nbody ? (int64_t)w * body : (int64_t)nbody * body + (int64_t)(w - nbody) * tail;
    }
    __host__ __device__ int32_t elems(int32_t w) const { return w < nbody ? body : tail; }
    __host__ __device__ int32_t chunk_of(int64_t e) const {
        const int64_t edge = (int64_t)nbody * body;
        return e < edge ? (int32_t)(e / body) : nbody + (int32_t)((e - edge) / tail);
    }
};

inline size_t workspace_bytes_for(int32_t nchunks) {
    // carry_head[nchunks] + carry_tail[nchunks] (double) + carry_info[nchunks] (int4)
    size_t b = (size_t)nchunks * (8 + 8 + 16);
    return (b + 255) & ~(size_t)255;
}

hipError_t launch_column_sums(const double* d_x, const int32_t* d_p, int32_t ncol, int32_t nnz,
                              double* d_out, const LaunchPlan& plan, void* d_workspace,
                              double divisor, bool means, hipStream_t stream, int op = kOpSum,
                              const int32_t* rows_i = nullptr, const uint32_t* row_bitmap = nullptr,
                              int32_t bitmap_words = 0, const int2* plan_rec = nullptr,
                              const int32_t* run_if = nullptr, uint32_t* fold_ticket = nullptr);
// fold_ticket: a zeroed device word owned by the launching stream (fold_ticket_address): a plain column-sum call that is one
// round of waves (plan.short_pipeline) then runs its fix-up inside the main launch -- ONE kernel -- with identical bits
hipError_t fold_ticket_address(int slot, uint32_t** out);
// n doubles from HBM into page-locked host memory (its device address) by a kernel, enqueued on `stream`
hipError_t launch_copy_f64(const double* d_src, double* dst, int64_t n, hipStream_t stream);

// true in a process forked from one that had already asked this library for a device (the HIP runtime does not survive a fork)
bool process_was_forked_after_gpu_use();
// rsp_column_sums_device without the entry's own planning (capi.hip): the library's one-shot paths use it
int column_sums_general(const double* d_x, const int32_t* d_p, int32_t ncol, int64_t nnz, double* d_out, void* d_ws,
                        size_t ws_bytes, hipStream_t stream);

// ---- what multigpu.cpp sees of a resident shard (struct rsp_csc is private to capi.hip) ----
struct CscView {
    int device;
    int32_t nrow, ncol;
    int64_t nnz;
    double* d_out;        // the handle's own output, ncol doubles in its device's HBM
    hipStream_t stream;   // the handle's own non-blocking stream
};
}  // namespace rsp
struct rsp_csc;
namespace rsp {
int csc_view(rsp_csc* h, CscView* v);
// column sums (means) of the shard enqueued on ITS stream into d_out (nullptr: its own output); the shard's device must be
// the calling thread's current one; nothing waits
int csc_enqueue_columns(rsp_csc* h, bool means, double* d_out);
// the shard's partial row sums (nrow doubles) enqueued on its stream into d_out in its device's HBM; its device must be current
int csc_enqueue_rows(rsp_csc* h, double* d_out);
// a shard over x / i / p that already live in `device`'s HBM and stay the caller's (never copied, never freed)
int csc_wrap_device(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t nrow, int32_t ncol, int64_t nnz,
                    int device, rsp_csc** handle);

// ---- device-side inspector (inspect_device.hip): the plans of inspect.hpp for offsets that live in HBM ----
constexpr int kInspectMaxBlocksColumns = 1024;   // blocks of the pass over p[] (grid-stride): one partial record each
constexpr int kInspectMaxBlocksChunks = 256;     // blocks of the pass over the lean chunks
typedef inspect::Stats PlanStats;   // (inspect.hpp: shared with the host restatement of the same pass)
struct DeviceInspectLayout {   // the plan memory of a device-made plan (byte offsets, 256-aligned); sizes depend on ncol, nnz and the settings only
    size_t part1_off, part2_off, rec_off, first_off, hdr_off, bytes;   // part1 / part2: the blocks' partial statistics
    bool try_lean;             // mean column length <= kLeanMaxColumn: otherwise some column is too long for the lean form
    int32_t lean_rows, lean_chunks;
    int32_t lean_capacity, lean_capacity_stride;   // columns per chunk the image has room for, and that stride in dwords
};
hipError_t launch_inspect_device(const int32_t* d_p, int32_t ncol, int32_t nnz, const LaunchPlan& grid,
                                 const DeviceInspectLayout& L, void* d_mem, PlanStats* stats_out, hipStream_t stream);

// hand-written exclusive prefix sum of 32-bit counts (scan.hip): out[k] = initial + in[0] + ... + in[k - 1]; out may be in;
// out2: an optional second copy of the result; run_if: an optional device word -- 0 when the launches run = they do nothing
size_t exclusive_scan_temp_bytes(int64_t n);
hipError_t launch_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t initial, void* temp,
                                     size_t temp_bytes, hipStream_t stream, const int32_t* run_if = nullptr,
                                     int32_t* out2 = nullptr);

// the main kernel's access shape without any column work (rsp_debug_read_ceiling_device)
hipError_t launch_read_ceiling(const double* d_x, int32_t nnz, const LaunchPlan& plan, double* d_sink,
                               hipStream_t stream);

// Row-restricted sums over more than 2^20 rows, slice-major form (colsums_rowslices.hip): a workgroup keeps ONE
// slice of 2^20 rows of the bitmap in LDS and walks its group of columns slice by slice.
constexpr int kSliceRowsShift = 20;        // rows per slice: 128 KB of bitmap
constexpr int kSliceMaxGroup = 2048;       // columns per workgroup: a cursor (4 B) and a sum (8 B) each beside the bitmap
constexpr int kSliceMinColumns = 13312;    // fewer columns leave wavefronts of the 256 workgroups without a batch (1e9 entries over 1e7 rows, slices against L2 probes: 1e4 columns 2.68 / 2.18 ms, 1.6e4 2.29 / 2.62, 2e4 2.16 / 2.88; round 4's edge sweep moved this from 16384)
constexpr int kSliceMinSegment = 32;       // mean entries per (column, slice) from which the form is selected
constexpr int kSliceMinEntriesPerPass = 32768;   // entries of a column group per slice (393 KB of x and i against the 128 KB of bitmap copied for them)
constexpr int kSliceCus = 256;             // MI355X: one workgroup per CU, groups sized for whole rounds of them
constexpr int kSliceMaxColumnFactor = 16;  // guard: a column longer than 16 x the mean (+ 4096) goes back to the general kernel
struct RowSlicesPlan {
    int32_t nslices, group, ngroups, max_col;
    int64_t max_group;
};
bool rowslices_applicable(int32_t nrow, int32_t ncol, int64_t nnz, bool force, RowSlicesPlan* out);
// zeroes *d_flag and sets it when the matrix has a column / column group the slice form should not take
hipError_t launch_rowslices_guard(const int32_t* d_p, int32_t ncol, const RowSlicesPlan& sp, int32_t* d_flag,
                                  hipStream_t stream);
// runs unless *d_skip_if != 0 (d_skip_if may be null)
hipError_t launch_column_sums_rowslices(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t ncol,
                                        int32_t nnz, const uint32_t* d_bitmap, int32_t bitmap_words, bool complement,
                                        const RowSlicesPlan& sp, double* d_out, const int32_t* d_skip_if,
                                        hipStream_t stream);

// hipFuncSetAttribute(..MaxDynamicSharedMemorySize..) once per (kernel, device): the attribute belongs to the
// device's copy of the function, so a process that uses a second device has to raise it there as well.
struct DynamicLdsLimit {
    std::atomic<uint64_t> done{0};   // one bit per device ordinal below 64 (benign if two threads both raise it)
    hipError_t ensure(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1)) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or((uint64_t)1 << dev, std::memory_order_relaxed);
        return e;
    }
};

// Workspace layout of the row-wise path (rowsums.hip); offsets in bytes, 256-aligned.
// persistent = the entries grouped by 16384-row block (values, row indices; not in the direct form), the first
// slot of every block and the parts' sums: what a handle keeps between calls.  scratch = the (block, supertile)
// count table + scan temp, and in the two-level form (mode 3, more than 832 blocks) the intermediate copy grouped
// by bucket of 512 blocks with the buckets' first slots; free again once the build has run.
struct RowSumsLayout {
    int mode;   // 2 = one partition pass (or none: direct), 3 = two passes
    size_t vals_off, rows_off, boff_off, partial_off, persistent_bytes;
    size_t table_off, temp_off, temp_bytes, mid_vals_off, mid_rows_off, bucket_off, scratch_bytes;
    int32_t shift, nblocks, nsuper, nsplit, nbuckets;   // nsplit: accumulate workgroups per row block
    int32_t sub, ncoarse;   // the partition pass groups by coarse block of 2^sub row blocks (ncoarse of them)
    bool aligned;           // the pass writes whole groups of 8 entries only; regions padded with entries of no row
    int64_t slots;          // capacity of the regrouped copy in entries (nnz, or nnz + the padding)
    bool direct;   // up to 4 row blocks: no regrouping, the accumulate pass reads the caller's x / i
    bool rows16;   // the regrouped copy keeps a row as 16 bits RELATIVE TO ITS ROW BLOCK (0xffff: no row): one-level regrouping by the row blocks themselves, where a region belongs to one block -- 10 instead of 12 B per entry written by the partition pass and read by every accumulate pass (round 6)
    int64_t super_elems;
};
hipError_t plan_row_sums(int32_t nrow, int64_t nnz, size_t colsums_ws_bytes, bool keep_row_form, RowSumsLayout* L);
hipError_t launch_row_build(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz,
                            const RowSumsLayout& L, void* persist, void* scratch, hipStream_t stream);
hipError_t launch_row_reduce(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz,
                             const RowSumsLayout& L, void* persist, double* d_out,
                             double divisor, bool means, const LaunchPlan& colsums_plan, hipStream_t stream);

// columns planned form (every column long and of similar length): one workgroup of `waves` (4 / 8 / 16) wavefronts per column
constexpr int kColumnsMinLen = 2048;       // shortest column the form takes (16 rows of 128 entries)
constexpr int kColumnsMinLenTwoWaves = 512;   // ... or, with two wavefronts per column, this short in matrices of up to
constexpr int64_t kColumnsTwoWavesMaxNnz = 250000000;   // ... this many entries
constexpr int kColumnsMaxLen = 1 << 22;    // longest (its bytes stay far below a buffer descriptor's 2^31)
constexpr int kColumnsMaxOverMean = 4;     // no column longer than this many times the mean (one workgroup walks it)
constexpr int kColumnsMinColumns = 128;
constexpr int kColumnsFewMaxLen = 45056;   // fewer columns than that take the form too while the longest column has at most this many entries + nnz / 192: one workgroup streams a column alone (3.3 us + 0.14 us per 1000 entries) against the general kernels' 8.5 us + 1.1 us per 1e6 entries of the whole call (8..127 columns of 3e3..3e5 entries, profiles/r04_form_edges.json)
// stale != nullptr: the guarded build (offsets clamped to [0, nnz]; a column length outside [len_lo, len_hi] sets *stale)
hipError_t launch_column_sums_columns(const double* d_x, const int32_t* d_p, int32_t ncol, int32_t waves,
                                      double* d_out, double divisor, bool means, hipStream_t stream, int32_t nnz = 0,
                                      int32_t len_lo = 0, int32_t len_hi = 0, int32_t* stale = nullptr);

// Segments form of the row sums (rowsums.hip): behind a handle whose columns' rows ascend; no regrouped copy
struct RowSegmentsLayout {
    int32_t shift, nblocks, nsplit;
    size_t table_off, cuts_off, flag_off, partial_off, bytes;
};
bool row_segments_applicable(int32_t nrow, int32_t ncol, int64_t nnz, bool force);
hipError_t plan_row_segments(int32_t nrow, int32_t ncol, int64_t nnz, RowSegmentsLayout* L);
// *d_flag = 1 if some column's rows do not ascend (0 otherwise)
hipError_t launch_rows_sorted_check(const int32_t* d_i, const int32_t* d_p, int32_t ncol, int64_t nnz,
                                    int32_t* d_flag, hipStream_t stream);
hipError_t launch_row_segments_build(const int32_t* d_i, const int32_t* d_p, int32_t ncol, int64_t nnz,
                                     const RowSegmentsLayout& L, void* persist, hipStream_t stream);
hipError_t launch_row_segments_reduce(const double* d_x, const int32_t* d_i, int32_t nrow, int32_t ncol,
                                      const RowSegmentsLayout& L, void* persist, double* d_out, double divisor,
                                      bool means, hipStream_t stream);

// d_p and stale given: the self-validating build (every column's image offsets compared with d_p; a column that differs is
// summed again straight from x and *stale is set)
hipError_t launch_column_sums_lean(const double* d_x, int32_t nnz, const int2* d_hdr, const uint32_t* d_offs,
                                   int32_t stride_dwords, int32_t nchunks, int32_t rows, double* d_out, double divisor,
                                   bool means, hipStream_t stream, const int32_t* d_p = nullptr, int32_t ncol = 0,
                                   int32_t* stale = nullptr);

// out[j] = ((part_0[j] + part_1[j]) + ...) + part_{nparts-1}[j] (+ 0.0, / divisor): the shards' partial row sums
// added in shard order.  part_k = parts + k * stride, except part own_idx = own (own_idx < 0: none).
hipError_t launch_add_partials(const double* parts, int32_t nparts, int64_t stride, const double* own,
                               int32_t own_idx, int64_t n, double* out, double divisor, bool means,
                               hipStream_t stream);

// Matrix::crossprod on the device (crossprod.hip): dense ncol x ncol, column-major.
struct CrossprodLayout {   // workspace of the row-major path
    size_t rp_off, cursor_off, rc_off, rx_off, temp_off, temp_bytes, total_bytes;
    int32_t nsplit, width;   // slices of a result column and their width (crossprod_split)
    // tall form (few long columns; matrix cores, not bit-identical): workgroups, row panels per workgroup,
    // column tiles; their results and the "x holds a non-finite value" flag in the workspace
    bool tall;
    int32_t ngroups, panels_per_group, ntiles;
    size_t partial_off, flag_off;
    // 16 tiles: the tall form finds its 32-row panels through a table T[panel][column] (crossprod.hip) in the workspace
    bool panel_table;
    int32_t panel_rows;   // 32; 16 at 32 tiles
    int64_t npanels;
    size_t table_off, has_off;
};
hipError_t plan_crossprod(int32_t nrow, int32_t ncol, int64_t nnz, bool exact, CrossprodLayout* L);
void crossprod_split(int32_t nrow, int32_t ncol, int64_t nnz, int32_t* nsplit, int32_t* width);
constexpr int kXpAll = 0, kXpTallOnly = 1, kXpExactOnly = 2;   // which part of the work launch_crossprod_rows enqueues
hipError_t launch_crossprod_rows(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t nrow,
                                 int32_t ncol, int64_t nnz, double* d_out, const CrossprodLayout& L, void* ws,
                                 hipStream_t stream, int part = kXpAll);
hipError_t launch_crossprod(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t ncol,
                            double* d_out, hipStream_t stream);

hipError_t launch_gen_row_indices(int32_t* d_i, const int32_t* d_p, int32_t nrow, int32_t ncol,
                                  uint64_t seed, hipStream_t stream);
hipError_t launch_gen_values(double* d_x, int64_t n, uint64_t seed, uint64_t first_idx, int kind,
                             hipStream_t stream);

}  // namespace rsp
#endif

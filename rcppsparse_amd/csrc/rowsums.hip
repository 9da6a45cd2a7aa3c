// rowsums.hip -- Matrix::rowSums / rowMeans on the device ("next" row f1 of SURVEY.md 8f).
//
// Reference inst/include/RcppSparse.h:138-144 scatters sums(i[j]) += x[j] while walking the
// columns.  On the device that is a reduction by key with 1e7 keys and no locality in the key.
// All hand-written; forms chosen in plan_row_sums by the number of 16384-row blocks (what one CU's LDS holds as sums):
//
//   direct (up to 4 row blocks, 65536 rows): nothing is regrouped.  Steps 4 and 5 below run on the caller's
//     x / i, every block's workgroups scanning all entries and adding those of their block: 12 B/nnz per block,
//     no workspace beyond the parts' sums (5e8 nnz: 1.0 / 1.4 / 2.9 / 2.8 ms for 1 / 2 / 3 / 4 blocks; regrouping
//     first costs 4.2-4.9 ms, and from 5 blocks on is the faster way: 4.4 against 4.8 ms).
//
//   tile partition (5 to 832 row blocks = up to 1.36e7 rows): the entries are regrouped by ROW BLOCK in ONE pass,
//     then added up block by block.
//       1. rows_tile_histogram_kernel   entries per (block, supertile)              reads  4 B/nnz
//       2. one exclusive scan over that table (scan.hip; a few MB) = first output slot of every pair
//       3. rows_tile_partition_kernel   a workgroup walks its supertile in tiles of 22528 entries (22
//          per thread, in registers): ranks them per block with wave-private LDS counters, SORTS THE
//          TILE BY BLOCK THROUGH LDS (8192 positions per round), and writes every block's run to its
//          cursor -- consecutive lanes store consecutive slots, so the ~600 output streams leave as
//          runs of ~37 entries instead of single 8-byte stores     reads 12 B/nnz, writes 12 B/nnz
//       4. rows_tile_accumulate_kernel  16-wave workgroups, `nsplit` per block: the block's sums live
//          in 128 KB of LDS; fifteen wavefronts stage entries, the sixteenth adds the previous step's
//          with ds_add_f64, in slot order, from (byte offset, value) pairs the stagers prepared
//                                                                   reads 12 B/nnz, writes 8 B/row
//       5. rows_combine_parts_kernel    (nsplit > 1) adds the parts of every row in part order
//     About 40 B/nnz of traffic and a workspace of 12 B/nnz + the count table.  An entry's slot is a
//     function of the data alone (no global atomics, wave-private counters combined in a fixed
//     order), and a row's terms are added by one wavefront in slot order: bit-stable run to run.
//
//     Round 3, 512-640 blocks (8.4e6-1.05e7 rows, C3): step 3 runs in its QUEUE form (rows_tile_partition_queue_kernel)
//     -- only whole aligned groups of 16 entries leave for HBM, through one queue per block in LDS -- because the pass
//     is bound by its write side (profiles/r03_rowsums_write_side.json); clustered row indices (a cell of the count
//     table above 96 entries per tile) take the staged form above, which stands by on the same table.
//
//   coarse blocks (round 3; up to 8 x 832 row blocks = 1.09e8 rows): the same single pass regroups by COARSE block
//     of 2 / 4 / 8 row blocks, and in step 4 every row block scans its coarse block's entries for its own (12 B/nnz
//     x 2 / 4 / 8 there).  1e9 entries: 12.9 ms at 2e7 rows, 17.3 ms at 4e7 (a library radix sort by block: 18.8 ms
//     at 2e7).
//
//   two levels (round 3; more rows still): a first pass regroups by BUCKET of 512 row blocks into an intermediate
//     copy, a second regroups every bucket by its blocks (the same kernels on the bucket's range): 21.2 ms for 1e9
//     entries in 1.2e8 rows, workspace 24 B/nnz.
//
// A handle (rsp_csc_row_sums) keeps the regrouped copy and repeats steps 4 and 5 only (12 B/nnz per call) -- or, where
// its columns are long and their rows ascend, regroups nothing at all (segments form, at the end of this file).
// Rounds 1-2 sorted with rocPRIM here (by 4096-row block above 1.36e7 rows, fully by row behind the handle); the
// exclusive scan of the count table was the last library call (round 4: scan.hip).
// All forms are deterministic and within the usual 1e-12 * sum|x| of the reference's order.
// Round-2 history (profiles/r02_rowsums.md): partitioning WITHOUT sorting each tile in LDS first
// (every lane storing its entry straight to its block's cursor) took 47 ms for the scatter alone.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// tile partition
constexpr int kPartShift = 14;            // 16384 rows per block: 128 KB of LDS sums per workgroup
constexpr int kPartMaxBlocks = 832;       // LDS counters / cursors of the partition kernel (with the stage: < 160 KB)
constexpr int kPartBucketShift = kPartShift + 9;   // two-level form: a bucket of the first pass = 512 blocks = 8.4e6 rows
constexpr int kPartMaxSub = 3;            // up to 2^3 row blocks per coarse block before the two-level form takes over
constexpr int kPartTwoLevelBlocks = 512;  // blocks either pass of the two-level form separates (first pass: <= 256 buckets)
constexpr int kPartThreads = 1024;        // partition workgroup: 16 wavefronts
constexpr int kPartWaves = kPartThreads / 64;
constexpr int kPartPerThread = 22;        // entries a thread holds in registers while its tile is ranked (24 spills)
constexpr int kTileElems = kPartThreads * kPartPerThread;   // 22528 entries ranked together ...
constexpr int kStageElems = 8192;         // ... and sorted through LDS 8192 positions at a time
// queue form of the partition pass (round 3): only whole, aligned groups of kQueueGroup entries (128 B of values, 64 B
// of row indices) leave for HBM; every block has a queue of one group in LDS, which takes the place of the stage
constexpr int kQueueGroup = 16;
constexpr int kQueueMinBlocks = 512;      // below: a tile's run per block is long enough for the staged form's plain write-out
constexpr int kQueueMaxBlocks = 640;      // 232 B of LDS per block: queue, cursor, tile count, 16 packed counters
constexpr int kQueuePerThread = 16;       // entries a thread holds per tile of the queue form (22 spill there: 81 registers)
constexpr int kQueueTileElems = kPartThreads * kQueuePerThread;   // 16384: run length no longer matters, whole groups leave
constexpr int kQueueCellPerTile = 96;     // a (block, supertile) cell above this many entries per tile of the supertile: staged form
constexpr int kTilesPerSuper = 7;         // a workgroup's supertile: at most 157 696 entries
constexpr size_t kCountTableMaxBytes = 64u << 20;
constexpr int kAccThreads = 1024;         // accumulate workgroup: 15 wavefronts stage the entries, ONE adds them
constexpr int kAccStagers = kAccThreads - 64;
constexpr int kAccDepth = 8;              // steps of entries a staging thread has in flight (8 x 11.5 KB per CU)
constexpr int kAccMaxSplit = 1024;        // workgroups that may share one row block
#ifndef RSP_DIRECT_XCD_RUNS
#define RSP_DIRECT_XCD_RUNS 1
#endif
constexpr bool kDirectXcdRuns = RSP_DIRECT_XCD_RUNS != 0;   // direct form: the readers of a part run together on one XCD
constexpr int kDirectMaxBlocks = 4;       // up to here the accumulate pass scans the caller's x / i once per block instead

// ---------------------------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------------------------
// Workgroups per row block in the accumulate pass.  One per block leaves the last round of a grid of
// 611 blocks on 256 CUs 61 % empty (and a 62-block matrix on a quarter of the chip); the blocks'
// entries are split into equal parts instead, each part summed by its own workgroup and the parts
// added up in a fixed order afterwards.  Cost model: rounds x bytes a workgroup moves (its share of the
// `stream_entries` entries a block's workgroups scan between them, plus its block of sums written and read back).
static int accumulate_split(int32_t nblocks, int64_t stream_entries, int64_t rows_per_block) {
    static int ncu = 0;   // (benign if two threads both fill it in)
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            ncu = n;
        else {
            (void)hipGetLastError();
            return 1;   // no device to plan for (workspace queries on a host without one)
        }
    }
    static const int candidates[] = {1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024};
    int best = 1;
    double best_cost = 0;
    for (int ns : candidates) {
        if (ns > kAccMaxSplit) break;
        const int64_t units = (int64_t)nblocks * ns;
        const int64_t rounds = (units + ncu - 1) / ncu;
        const double unit_bytes = 12.0 * (double)stream_entries / (double)ns + (ns > 1 ? 24.0 : 8.0) * (double)rows_per_block;
        const double cost = (double)rounds * unit_bytes;
        if (ns == 1 || cost < best_cost * 0.97) {   // (a split has to pay for itself)
            best = ns;
            best_cost = cost;
        }
    }
    return best;
}

hipError_t plan_row_sums(int32_t nrow, int64_t nnz, size_t colsums_ws_bytes, bool keep_row_form,
                         RowSumsLayout* L) {
    (void)colsums_ws_bytes;
    (void)keep_row_form;   // (a handle keeps the regrouped copy -- `persistent` -- and repeats the accumulate pass only)
    memset(L, 0, sizeof(*L));
    size_t off = 0;

    const int64_t part_blocks = ((int64_t)nrow + (1 << kPartShift) - 1) >> kPartShift;
    L->shift = kPartShift;
    L->nblocks = (int32_t)(part_blocks > 0 ? part_blocks : 1);
    // More row blocks than one partition pass separates (832 = 1.36e7 rows): regroup by COARSE block of 2, 4 or 8
    // row blocks -- still one pass -- and let every row block scan its coarse block's entries for its own in the
    // accumulate pass (12 B/nnz x 2, 4, 8 there: 2e7 rows 13 ms where two partition passes take 21 and a radix sort
    // by block took 19).  Beyond 8 (1.09e8 rows) the two-level form takes over.
    L->sub = 0;
    while (L->sub < kPartMaxSub && ((L->nblocks + (1 << L->sub) - 1) >> L->sub) > kPartMaxBlocks) ++L->sub;
    // RSP_ROWS_SUB=1 / 2 / 3: regroup by coarse blocks of at least 2 / 4 / 8 row blocks also where one pass could separate
    // the row blocks themselves (round 4's last structural attempt on the one-shot call at C3: fewer, longer output
    // streams in the partition pass against 2 / 4 / 8 x the accumulate pass's reads; profiles/r04_rowsums.md)
    static const int forced_sub = [] {
        const char* e = getenv("RSP_ROWS_SUB");
        const int v = e ? atoi(e) : 0;
        return v < 0 ? 0 : (v > kPartMaxSub ? kPartMaxSub : v);
    }();
    if (forced_sub > L->sub && L->nblocks > kDirectMaxBlocks) L->sub = forced_sub;
    L->ncoarse = (L->nblocks + (1 << L->sub) - 1) >> L->sub;
    L->mode = L->ncoarse <= kPartMaxBlocks ? 2 : 3;
    if (L->mode == 3) {
        L->sub = 0;
        L->ncoarse = L->nblocks;
    }
    // supertile: whole tiles, small enough for ~1000 workgroups when the matrix allows, large enough
    // for the count table to fit its budget
    int64_t tiles = nnz / ((int64_t)1024 * kTileElems);
    tiles = tiles < 1 ? 1 : tiles > kTilesPerSuper ? kTilesPerSuper : tiles;
    int64_t super = (int64_t)kTileElems * tiles;
    const int32_t table_blocks = L->mode == 2 ? L->ncoarse : kPartTwoLevelBlocks;   // blocks one pass separates
    while (((nnz + super - 1) / super) * table_blocks * 4 > (int64_t)kCountTableMaxBytes) super *= 2;
    L->super_elems = super;
    L->nsuper = (int32_t)((nnz + super - 1) / super);
    const int64_t rows_here = nrow < (1 << kPartShift) ? (nrow > 0 ? nrow : 1) : (1 << kPartShift);
    // Few blocks: no regrouping at all.  A block's workgroups scan the caller's x / i as they are and add
    // the entries of their block (12 B/nnz per block instead of ~40 B/nnz and the partition pass's LDS work;
    // 5e8 nnz: 1.0 ms for one block, 2.8 ms for four; the partition form takes 4.2-4.9 ms at 2-61 blocks).
    L->direct = L->nblocks <= kDirectMaxBlocks;
    L->nsplit = accumulate_split(L->nblocks, L->direct ? nnz : nnz / L->ncoarse, rows_here);
    // Many blocks (more than one pass's LDS counters hold: 832 = 1.36e7 rows): two levels.  The first pass regroups
    // by BUCKET of 512 blocks (8.4e6 rows; at most 256 buckets for 2^31 rows) into an intermediate copy, the second
    // regroups every bucket by its blocks -- the same two kernels, run on the bucket's range of the intermediate.
    L->nbuckets = L->mode == 3 ? (int32_t)(((int64_t)nrow + ((int64_t)1 << kPartBucketShift) - 1) >> kPartBucketShift) : 0;
    const size_t table_entries = (size_t)L->nsuper * (size_t)table_blocks + 1;   // + the total
    const size_t temp = exclusive_scan_temp_bytes((int64_t)table_entries);   // (scan.hip: hand-written since round 4)
    // queue form of the partition pass (whole aligned groups of 16 entries only): every (block, supertile) region is
    // padded to whole groups, so the regrouped copy has up to 15 slots more per region; kept to what 32-bit slots hold
    static const bool queue_allowed = [] {   // RSP_ROWS_QUEUE=0: the round-2 write-out (A/B measurements)
        const char* e = getenv("RSP_ROWS_QUEUE");
        return !(e && atoi(e) == 0);
    }();
    const int64_t padded = nnz + (int64_t)(kQueueGroup - 1) * L->ncoarse * L->nsuper;
    // (measured, 1e9 entries, queue against staged form: 611 blocks 7.3-7.4 against 8.2 ms on the slower devices of
    // the pool, 9.32 against 9.57 ms per call on a faster one; 305 blocks 9.24 against 8.89 ms: with fewer blocks a
    // tile's runs are long enough as they are, and the queues need more rounds)
    L->aligned = queue_allowed && L->mode == 2 && !L->direct && L->ncoarse >= kQueueMinBlocks &&
                 L->ncoarse <= kQueueMaxBlocks && padded <= 0x7fffffffll;
    L->slots = L->aligned ? padded : nnz;
    // one-level regrouping by the row blocks themselves: every region of the copy belongs to ONE block, so a row is kept
    // as its 14 bits inside the block (0xffff: an entry of no row) -- 2 bytes instead of 4 (RSP_ROWS16=0: A/B)
    static const bool rows16_allowed = [] {
        const char* e = getenv("RSP_ROWS16");
        return !(e && atoi(e) == 0);
    }();
    L->rows16 = rows16_allowed && L->mode == 2 && !L->direct && L->sub == 0 && kPartShift <= 15;
    if (!L->direct) {
        L->vals_off = off;  off = align_up(off + (size_t)L->slots * 8, 256);       // x grouped by row block
        L->rows_off = off;  off = align_up(off + (size_t)L->slots * (L->rows16 ? 2 : 4), 256);   // their row indices
    }
    L->boff_off = off;  off = align_up(off + ((size_t)L->ncoarse + 1) * 4, 256);   // first slot of every (coarse) block
    L->partial_off = off;                                                          // sums per (part, row)
    if (L->nsplit > 1) off = align_up(off + (size_t)L->nsplit * (size_t)(nrow > 0 ? nrow : 0) * 8, 256);
    L->persistent_bytes = off;
    off = 0;
    L->table_off = off; off = align_up(off + (table_entries + 1) * 4, 256);   // (+ the clustered-rows flag of the queue form)
    L->temp_off = off;  off = align_up(off + temp, 256);
    L->temp_bytes = temp;
    if (L->mode == 3) {   // the intermediate copy (grouped by bucket) and the buckets' first slots
        L->mid_vals_off = off;  off = align_up(off + (size_t)nnz * 8, 256);
        L->mid_rows_off = off;  off = align_up(off + (size_t)nnz * 4, 256);
        L->bucket_off = off;    off = align_up(off + ((size_t)L->nbuckets + 1) * 4, 256);
    }
    L->scratch_bytes = off;
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// LDS adds
// ---------------------------------------------------------------------------------------------
// sums[r] += v as one LDS instruction (an IEEE double add performed by the LDS unit).  A wave's LDS
// instructions execute in issue order; lanes of one instruction that hit the same row are serialised
// by the hardware, always the same way.
__device__ __forceinline__ void lds_add_f64(double* a, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)a, v);
}

// ---------------------------------------------------------------------------------------------
// tile partition: histogram, partition, accumulate
// ---------------------------------------------------------------------------------------------
// Workgroup barrier for LDS hand-offs only.  __syncthreads() also waits for every outstanding global
// load and store of the wave (s_waitcnt vmcnt(0)); in the two kernels below that would expose the
// latency of the loads just issued for the NEXT step and of the stores of the step just written out,
// once per step (measured on the partition pass: 14 of 18 us per tile).  Loaded registers are still
// waited for where they are used (the compiler counts those), and a store has read its registers
// when it has issued, so neither needs the full drain.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A wave-uniform pointer, moved to scalar registers.
__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (void*)(((uint64_t)hi << 32) | lo);
}

// 1. entries per (row block, supertile).  The table is [block][supertile], so that ONE flat exclusive
//    scan yields for every pair the number of entries in earlier blocks plus those of the same block
//    in earlier supertiles: its first output slot.  (Row indices outside [0, nrow) -- not a valid
//    dgCMatrix -- are neither counted here nor moved below.)
//    `seg` (two-level form): the pass works on entries [seg[0], seg[1]) only -- one bucket of the first level --
//    and on rows [row_base, row_base + nrow); nullptr = all nnz entries.  The grid is sized for the worst case
//    (the host does not know the bucket sizes), supertiles past the segment's end count nothing.
__global__ __launch_bounds__(kPartThreads) void rows_tile_histogram_kernel(
    const int32_t* __restrict__ ri, int64_t nnz, int32_t nrow, int32_t shift, int32_t nblocks,
    int64_t super_elems, int32_t nsuper, int32_t* __restrict__ table, const int32_t* __restrict__ seg,
    int32_t row_base, int32_t group, int32_t* __restrict__ skew_flag, int32_t cell_limit) {
    extern __shared__ int32_t s_hist[];
    const int tid = threadIdx.x;
    const int s = blockIdx.x;
    for (int b = tid; b < nblocks; b += kPartThreads) s_hist[b] = 0;
    __syncthreads();
    const int64_t seg0 = seg ? seg[0] : 0, seg1 = seg ? seg[1] : nnz;
    int64_t e0 = seg0 + (int64_t)s * super_elems;
    e0 = e0 < seg1 ? e0 : seg1;
    const int64_t e1 = e0 + super_elems < seg1 ? e0 + super_elems : seg1;
    // 16 bytes per lane and load (1 KB per wave instruction: 4-byte loads reach about two thirds of that rate), two
    // loads in flight per thread; a buffer resource so that the last, partial quad reads zeros instead of faulting
    const int32_t len = __builtin_amdgcn_readfirstlane((int32_t)(e1 - e0));
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ri + e0), 0, len * 4, 0x00020000);
    typedef int32_t i4 __attribute__((ext_vector_type(4)));
    for (int32_t e = 0; e < len; e += 8 * kPartThreads) {
        i4 q[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            q[k] = __builtin_bit_cast(i4, __builtin_amdgcn_raw_buffer_load_b128(res, tid * 16, (e + k * 4 * kPartThreads) * 4, 2));
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int32_t j = e + k * 4 * kPartThreads + tid * 4 + c;
                const uint32_t r = (uint32_t)q[k][c] - (uint32_t)row_base;
                if (j < len && r < (uint32_t)nrow) atomicAdd(&s_hist[r >> shift], 1);
            }
    }
    __syncthreads();
    // (queue form of the partition pass: every (block, supertile) region holds whole groups of `group` slots)
    for (int b = tid; b < nblocks; b += kPartThreads) {
        table[(size_t)b * nsuper + s] = (s_hist[b] + group - 1) / group * group;
        // (queue form: a supertile that sends this many entries to ONE block could need dozens of queue rounds per tile
        // -- rows sorted or clustered; the whole call then takes the staged form instead)
        if (skew_flag && s_hist[b] > cell_limit) atomicOr(skew_flag, 1);
    }
}

// 3. the partition pass.  A tile of 22528 entries (22 per thread, in registers) is ranked at once, so a
//    block's run in it is ~37 entries at 600 blocks; the sorted order then passes through an LDS stage
//    of 8192 positions at a time ("rounds"), each written out with consecutive lanes on consecutive slots.
//    Run length is what the write side pays for: at 13-entry runs (8192-entry tiles) the pass took 8.4 ms
//    and PMC showed 1.74x the write requests of a coalesced copy, almost half of them 32-byte partials;
//    27-entry runs 6.8 ms, 54-entry runs 5.9 ms (profiles/r02_rowsums.md).  Per tile the kernel is
//    otherwise bound by its scattered LDS accesses (rank, place, stage: ~55 % of its cycles).
//    LDS: the stage (values, row indices), one cursor per block (next output slot of this supertile),
//    the tile's first sorted position per block, and one counter per (wavefront, block).
__global__ __launch_bounds__(kPartThreads) void rows_tile_partition_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, int64_t nnz, int32_t nrow, int32_t shift,
    int32_t nblocks, int64_t super_elems, int32_t nsuper, const int32_t* __restrict__ first_slot,
    double* __restrict__ px, int32_t* __restrict__ pr, const int32_t* __restrict__ seg, int32_t row_base,
    const int32_t* __restrict__ run_if, int32_t pad_group, uint16_t* __restrict__ pr16 = nullptr) {
    // (standing by for the queue form: runs only if the histogram pass found the row indices too clustered for it, and
    // then fills every region up to whole groups with entries of no row, as the queue form's layout expects)
    if (run_if && *run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    double* stage_x = (double*)s_raw;                                   // kStageElems
    int32_t* stage_r = (int32_t*)(stage_x + kStageElems);               // kStageElems
    int32_t* cursor = stage_r + kStageElems;                            // nblocks
    int32_t* tstart = cursor + nblocks;                                 // nblocks + 1
    int32_t* cnt = tstart + nblocks + 1;                                // kPartWaves x nblocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    // (two-level form: this launch regroups one bucket of the first level, entries [seg[0], seg[1]) of x / ri,
    // by rows relative to row_base, into the same range of px / pr)
    const int64_t seg0 = seg ? seg[0] : 0, seg1 = seg ? seg[1] : nnz;
    for (int b = tid; b < nblocks; b += kPartThreads) cursor[b] = first_slot[(size_t)b * nsuper + s] + (int32_t)seg0;
    int64_t e0 = seg0 + (int64_t)s * super_elems;
    e0 = e0 < seg1 ? e0 : seg1;
    const int64_t e1 = e0 + super_elems < seg1 ? e0 + super_elems : seg1;
    int32_t* mycnt = cnt + (size_t)wave * nblocks;
    // the supertile as two buffer resources: a load is then one VGPR offset (the thread) plus a scalar
    // offset (tile, k) -- no address registers, of which 48 loads would need 96 -- and reads past the
    // end return zero instead of faulting
    // (e0 is a 64-bit product, which the compiler computes in vector registers; a descriptor left there
    // costs a readfirstlane loop around every load)
    const int32_t len = __builtin_amdgcn_readfirstlane((int32_t)(e1 - e0));
    const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ri + e0), 0, len * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_x = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + e0), 0, len * 8, 0x00020000);
    // this thread's entries of the current tile (coalesced across the workgroup; the row indices first,
    // they are needed first).  The NEXT tile's are requested into the same registers as soon as the
    // current ones have all gone to the stage, so they travel during the last round's write-out.
    int32_t r[kPartPerThread], pos[kPartPerThread];
    double v[kPartPerThread];
    auto fetch = [&](int32_t tile) {
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k) {
            r[k] = __builtin_amdgcn_raw_buffer_load_b32(res_r, tid * 4, (tile + k * kPartThreads) * 4, 2);
        }   // (nothing but loads here: whatever touched a loaded register would wait for it on the spot)
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k)   // (aux 2 = nt: read once)
            v[k] = __builtin_bit_cast(
                double, __builtin_amdgcn_raw_buffer_load_b64(res_x, tid * 8, (tile + k * kPartThreads) * 8, 2));
    };
    fetch(0);
    for (int32_t tile = 0; tile < len; tile += kTileElems) {
        for (int k = tid; k < kPartWaves * nblocks; k += kPartThreads) cnt[k] = 0;
        lds_barrier();   // (also: cursors initialised / updated, previous tile's last round read out of the stage)
        // rank of every entry among the entries of the same block handled by the same wavefront:
        // program order inside the wavefront, hardware order inside one LDS instruction -- both fixed
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k) {
            // (a load past the end returned 0: give it, like every row outside [row_base, row_base + nrow), the mark -1)
            const uint32_t rel = (uint32_t)r[k] - (uint32_t)row_base;
            const bool ok = tile + k * kPartThreads + tid < len && rel < (uint32_t)nrow;
            r[k] = ok ? r[k] : -1;
            pos[k] = ok ? atomicAdd(&mycnt[rel >> shift], 1) : -1;
        }
        lds_barrier();
        // per block: counts of the wavefronts -> exclusive prefix over the wavefronts, total into tstart
        for (int b = tid; b < nblocks; b += kPartThreads) {
            int run = 0;
#pragma unroll
            for (int w = 0; w < kPartWaves; ++w) {
                const int c = cnt[w * nblocks + b];
                cnt[w * nblocks + b] = run;
                run += c;
            }
            tstart[b] = run;
        }
        lds_barrier();
        // exclusive scan of the totals over the blocks (one wavefront; nblocks <= 64 lanes x 13)
        if (wave == 0) {
            const int per = (nblocks + 63) >> 6;
            const int b0 = lane * per;
            int sum = 0;
            for (int k = 0; k < per; ++k)
                if (b0 + k < nblocks) sum += tstart[b0 + k];
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            int run = incl - sum;
            for (int k = 0; k < per; ++k)
                if (b0 + k < nblocks) {
                    const int c = tstart[b0 + k];
                    tstart[b0 + k] = run;
                    run += c;
                }
            if (lane == 63) tstart[nblocks] = incl;
        }
        lds_barrier();
        // position of every entry in the tile's order by block
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k)
            if (pos[k] >= 0) {
                const int b = ((uint32_t)r[k] - (uint32_t)row_base) >> shift;
                pos[k] += tstart[b] + mycnt[b];
            }
        const int total = __builtin_amdgcn_readfirstlane(tstart[nblocks]);   // (uniform, and the compiler knows)
        auto round = [&](int base, bool last) {
            // this round's positions into the stage ...
#pragma unroll
            for (int k = 0; k < kPartPerThread; ++k)
                if ((uint32_t)(pos[k] - base) < (uint32_t)kStageElems) {
                    stage_x[pos[k] - base] = v[k];
                    stage_r[pos[k] - base] = r[k];
                }
            if (last) fetch(tile + kTileElems);   // (past the supertile: zeros, unused)
            lds_barrier();
            // ... and out: consecutive threads, consecutive positions, consecutive slots inside a block's run
            const int n = total - base < kStageElems ? total - base : kStageElems;
            for (int j = tid; j < n; j += kPartThreads) {
                const int32_t rr = stage_r[j];
                const int b = ((uint32_t)rr - (uint32_t)row_base) >> shift;
                const int32_t dest = cursor[b] + (base + j - tstart[b]);
                px[dest] = stage_x[j];
                if (pr16) pr16[dest] = (uint16_t)((uint32_t)rr & ((1u << shift) - 1u));   // (every staged entry has a row)
                else pr[dest] = rr;
            }
            lds_barrier();
        };
        // (the last round is a separate copy of the code: with the fetch inside the loop the compiler has
        // every round wait for "its" loads, and on the in-order counter that means for the previous round's
        // stores as well -- 8.3 ms instead of 6.9)
        int base = 0;
        for (; base + kStageElems < total; base += kStageElems) round(base, false);
        round(base, true);
        for (int b = tid; b < nblocks; b += kPartThreads) cursor[b] += tstart[b + 1] - tstart[b];
    }
    if (pad_group > 1) {
        lds_barrier();
        for (int b = tid; b < nblocks; b += kPartThreads) {
            const int32_t cur = cursor[b];
            const int over = cur & (pad_group - 1);
            if (over)
                for (int32_t d = cur; d < cur - over + pad_group; ++d) {
                    px[d] = 0.0;
                    if (pr16) pr16[d] = 0xffffu;
                    else pr[d] = -1;
                }
        }
    }
}

// 3b. the QUEUE form of the partition pass (round 3; up to kQueueMaxBlocks blocks, one-level regrouping).
//    What the pass above pays for is its write side (profiles/r03_rowsums_write_side.json: 7.45 ms as shipped in round
//    2, 3.30 ms without its global stores, 4.73 ms with the same bytes stored contiguously; the same bytes as aligned
//    pieces scattered over the array: 64-byte pieces 10.8 ms, 128-byte 6.7 ms, 256-byte 5.6 ms): unaligned runs of ~37
//    entries into 1222 streams per workgroup.  Here only WHOLE, ALIGNED GROUPS of 16 entries leave for HBM (one full
//    128-byte line of values, half a line of indices, each written exactly once): every block owns a queue of one group
//    in LDS; a tile's entries are ranked per block as above (wave-private counters, two 16-bit halves per word), given
//    their position in the block's stream, and go through the queues in rounds -- round r takes the entries whose
//    stream position falls into the r-th group from the queue's start, then every full queue is written out by 16
//    consecutive lanes; a trailing incomplete group simply stays queued for the next tile.  Regions of (block,
//    supertile) are whole groups (the histogram pass pads its counts); what is queued at the end of the supertile goes
//    out filled up with entries of no row (index -1), which the accumulate pass adds to nothing.  No stage: the
//    queues are the sort.  An entry's slot is still a function of the data alone.
__global__ __launch_bounds__(kPartThreads) void rows_tile_partition_queue_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, int64_t nnz, int32_t nrow, int32_t shift,
    int32_t nblocks, int64_t super_elems, int32_t nsuper, const int32_t* __restrict__ first_slot,
    double* __restrict__ px, int32_t* __restrict__ pr, const int32_t* __restrict__ skew_flag,
    uint16_t* __restrict__ pr16 = nullptr) {
    if (*skew_flag != 0) return;   // clustered row indices: the staged form (standing by on the same table) does the pass
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    double* qx = (double*)s_raw;                                        // nblocks x kQueueGroup
    int32_t* qr = (int32_t*)(qx + (size_t)nblocks * kQueueGroup);       // nblocks x kQueueGroup
    int32_t* cursor = qr + (size_t)nblocks * kQueueGroup;               // nblocks: next stream slot of this supertile
    int32_t* tcount = cursor + nblocks;                                 // nblocks: this tile's entries per block
    uint32_t* cnt = (uint32_t*)(tcount + nblocks);                      // (kPartWaves / 2) x nblocks, two wavefronts per word
    int32_t* s_rounds = (int32_t*)(cnt + (size_t)(kPartWaves / 2) * nblocks);   // kPartWaves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    constexpr int G = kQueueGroup;
    for (int b = tid; b < nblocks; b += kPartThreads) cursor[b] = first_slot[(size_t)b * nsuper + s];
    int64_t e0 = (int64_t)s * super_elems;
    e0 = e0 < nnz ? e0 : nnz;
    const int64_t e1 = e0 + super_elems < nnz ? e0 + super_elems : nnz;
    uint32_t* mycnt = cnt + (size_t)(wave >> 1) * nblocks;
    const bool odd = wave & 1;
    const int32_t len = __builtin_amdgcn_readfirstlane((int32_t)(e1 - e0));
    const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ri + e0), 0, len * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_x = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + e0), 0, len * 8, 0x00020000);
    int32_t r[kQueuePerThread], pos[kQueuePerThread];
    double v[kQueuePerThread];
    auto fetch = [&](int32_t tile) {
#pragma unroll
        for (int k = 0; k < kQueuePerThread; ++k)
            r[k] = __builtin_amdgcn_raw_buffer_load_b32(res_r, tid * 4, (tile + k * kPartThreads) * 4, 2);
#pragma unroll
        for (int k = 0; k < kQueuePerThread; ++k)
            v[k] = __builtin_bit_cast(
                double, __builtin_amdgcn_raw_buffer_load_b64(res_x, tid * 8, (tile + k * kPartThreads) * 8, 2));
    };
    fetch(0);
    for (int32_t tile = 0; tile < len; tile += kQueueTileElems) {
        for (int k = tid; k < (kPartWaves / 2) * nblocks; k += kPartThreads) cnt[k] = 0u;
        if (tid < kPartWaves) s_rounds[tid] = 0;
        lds_barrier();   // (also: cursors initialised / updated, the previous tile's last flush has read the queues)
        // rank of every entry among the entries of the same block handled by the same wavefront
#pragma unroll
        for (int k = 0; k < kQueuePerThread; ++k) {
            const bool ok = tile + k * kPartThreads + tid < len && (uint32_t)r[k] < (uint32_t)nrow;
            r[k] = ok ? r[k] : -1;   // (a load past the end returned 0; rows outside [0, nrow) belong to no block)
            if (ok) {
                const uint32_t old = atomicAdd(&mycnt[(uint32_t)r[k] >> shift], odd ? 0x10000u : 1u);
                pos[k] = (int32_t)(odd ? old >> 16 : old & 0xffffu);
            } else {
                pos[k] = -1;
            }
        }
        lds_barrier();
        // per block: exclusive prefix over the wavefronts (<= 22528: fits the 16-bit halves), the tile's count, and
        // how many rounds the block needs = groups its queue passes through
        {
            int need = 0;
            for (int b = tid; b < nblocks; b += kPartThreads) {
                uint32_t run = 0;
#pragma unroll
                for (int w2 = 0; w2 < kPartWaves / 2; ++w2) {
                    const uint32_t c = cnt[(size_t)w2 * nblocks + b];
                    const uint32_t lo = run;
                    run += c & 0xffffu;
                    const uint32_t hi = run;
                    run += c >> 16;
                    cnt[(size_t)w2 * nblocks + b] = lo | (hi << 16);
                }
                tcount[b] = (int32_t)run;
                const int nd = (int)(((cursor[b] & (G - 1)) + (int32_t)run + G - 1) / G);
                need = nd > need ? nd : need;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const int o = __shfl_xor(need, d, 64);
                need = o > need ? o : need;
            }
            if (lane == 0) s_rounds[wave] = need;
        }
        lds_barrier();
        // position of every entry in its block's stream, counted from the start of the group the queue holds
#pragma unroll
        for (int k = 0; k < kQueuePerThread; ++k)
            if (pos[k] >= 0) {
                const int b = (uint32_t)r[k] >> shift;
                const uint32_t pre = mycnt[b];
                const int32_t q = pos[k] + (int32_t)(odd ? pre >> 16 : pre & 0xffffu) + (cursor[b] & (G - 1));
                pos[k] = ((q >> 4) << 14) | (b * G + (q & (G - 1)));   // (round, queue slot): b * 16 + 15 < 2^14
            }
        int rounds = 0;
#pragma unroll
        for (int w = 0; w < kPartWaves; ++w) rounds = s_rounds[w] > rounds ? s_rounds[w] : rounds;
        rounds = __builtin_amdgcn_readfirstlane(rounds);
        auto round = [&](int rr, bool last) {
            // this round's entries into their queues ...
#pragma unroll
            for (int k = 0; k < kQueuePerThread; ++k)
                if ((pos[k] >> 14) == rr) {   // (-1 >> 14 = -1: never a round)
                    qx[pos[k] & 0x3fff] = v[k];
                    qr[pos[k] & 0x3fff] = r[k];
                }
            if (last) fetch(tile + kQueueTileElems);   // (past the supertile: zeros, unused)
            lds_barrier();
            // ... and every queue that is full now goes out: 16 consecutive lanes, one aligned group
            for (int item = tid; item < nblocks * G; item += kPartThreads) {
                const int b = item >> 4, l = item & (G - 1);
                const int32_t cur = cursor[b];
                const int fill = cur & (G - 1);
                if (fill + tcount[b] - rr * G >= G) {
                    const int32_t dest = cur - fill + rr * G + l;
                    px[dest] = qx[item];
                    if (pr16) pr16[dest] = (uint16_t)((uint32_t)qr[item] & ((1u << shift) - 1u));   // (a full group holds entries with rows only)
                    else pr[dest] = qr[item];
                }
            }
            lds_barrier();
        };
        int rr = 0;
        for (; rr + 1 < rounds; ++rr) round(rr, false);
        round(rr, true);   // (the last round is a separate copy of the code, like in the pass above; rounds == 0: nothing to place)
        for (int b = tid; b < nblocks; b += kPartThreads) cursor[b] += tcount[b];
    }
    // what is still queued at the end of the supertile: out with it, the group filled up with entries of no row
    lds_barrier();
    for (int item = tid; item < nblocks * G; item += kPartThreads) {
        const int b = item >> 4, l = item & (G - 1);
        const int32_t cur = cursor[b];
        const int fill = cur & (G - 1);
        if (fill > 0) {
            const int32_t dest = cur - fill + l;
            px[dest] = l < fill ? qx[item] : 0.0;
            if (pr16) pr16[dest] = l < fill ? (uint16_t)((uint32_t)qr[item] & ((1u << shift) - 1u)) : (uint16_t)0xffffu;
            else pr[dest] = l < fill ? qr[item] : -1;
        }
    }
}

// boff[b] = first slot of block b (= the scanned table's entry for supertile 0), boff[nblocks] = number of
// entries with a valid row index (the scan's last element)
//    (two-level form: the slots are relative to the bucket's first entry seg[0]; only the last bucket writes the
//    closing entry, which is then the end of the last block)
__global__ void rows_tile_offsets_kernel(const int32_t* __restrict__ first_slot, int32_t nblocks, int32_t nsuper,
                                         int32_t* __restrict__ boff, const int32_t* __restrict__ seg, int32_t close) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t base = seg ? seg[0] : 0;
    if (b < nblocks) boff[b] = base + first_slot[(size_t)b * nsuper];
    if (b == nblocks && close) boff[b] = base + first_slot[(size_t)nblocks * nsuper];
}

// 4. `nsplit` 16-wave workgroups per row block, each over an equal part of the block's entries.  Fifteen
//    wavefronts stage entries into one of two LDS buffers (coalesced loads, eight steps of them in
//    flight per thread); the sixteenth adds the buffer staged in the previous step, 64 entries per LDS
//    instruction, in slot order.  The adding wavefront's instruction stream is the critical path of the
//    pass, so the stagers leave it nothing to decide: they hand over (byte offset of the row's sum, value)
//    per entry, and it runs ds_read_b32, ds_read_b64, ds_add_f64 -- 45 instructions per 960 entries.
//    (Letting every wavefront pick "its" rows out of each step instead is 16 times the instructions:
//    9.5 ms; one buffer instead of two makes staging and adding alternate: 3.7 ms; now 2.0 ms on C3,
//    which is the read of the 12 GB.)
//    nsplit == 1: sums (or means) straight to `out`.  Otherwise each part's sums go to part_out + part * nrow
//    and rows_combine_parts_kernel adds the parts up.
template <bool MEANS, bool R16 = false>   // R16: the copy's rows are 16-bit, block-local (RowSumsLayout::rows16)
__global__ __launch_bounds__(kAccThreads) void rows_tile_accumulate_kernel(
    const double* __restrict__ px, const int32_t* __restrict__ pr, const int32_t* __restrict__ boff,
    int64_t direct_nnz, int32_t nrow, int32_t shift, int32_t sub, int32_t nsplit, double* __restrict__ out,
    double* __restrict__ part_out, double divisor, const uint16_t* __restrict__ pr16 = nullptr) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    double* sums = (double*)s_raw;                          // 1 << shift
    double* st_x = sums + ((size_t)1 << shift) + 64;        // 2 x kAccStagers  (64 spare slots behind the sums: where nothing-to-add goes)
    int32_t* st_r = (int32_t*)(st_x + 2 * kAccStagers);     // 2 x kAccStagers: byte offsets into sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x / nsplit, part = blockIdx.x - b * nsplit;
    // Direct form with several row blocks: the workgroups (block 0..nb-1, part k) scan the SAME entries.  Dealt out
    // block-major they would run rounds apart and each fetch its part from HBM again (4 blocks: 48 B/nnz); dealt out so
    // that they are 8 apart in the grid they run together AND on the same XCD (workgroups go round the 8 XCDs), and the
    // second to fourth reader of a part find it in that XCD's L2.  Runs of 8 parts x nb blocks; the last, incomplete run
    // of the grid is dealt part-major (together, if not on one XCD).
    if (kDirectXcdRuns && direct_nnz >= 0) {
        const int nb = gridDim.x / nsplit;
        if (nb > 1) {
            const int per_run = 8 * nb, run = blockIdx.x / per_run, in_run = blockIdx.x - run * per_run;
            if ((run + 1) * 8 <= nsplit) {
                part = run * 8 + (in_run & 7);
                b = in_run >> 3;
            } else {
                const int left = nsplit - run * 8;   // parts in the incomplete run (1..7)
                part = run * 8 + in_run % left;
                b = in_run / left;
            }
        }
    }
    const int rows_here = 1 << shift, mask = rows_here - 1;
    for (int r = tid; r < rows_here; r += kAccThreads) sums[r] = 0.0;
    // this workgroup's part of the block: whole steps, the same for every run
    // (direct form: every block scans all of x / i, which are then the caller's arrays)
    // (sub > 0: the entries are grouped by COARSE block of 2^sub row blocks -- matrices of more than 832 row blocks --
    // and each of its row blocks scans the coarse block's entries for its own, like the direct form does with all)
    const int cb = b >> sub;
    const int32_t s0 = direct_nnz >= 0 ? 0 : boff[cb], s1 = direct_nnz >= 0 ? (int32_t)direct_nnz : boff[cb + 1];
    const int32_t steps_all = (int32_t)(((int64_t)s1 - s0 + kAccStagers - 1) / kAccStagers);
    const int32_t steps_per = (steps_all + nsplit - 1) / nsplit;
    const int64_t u0_ = (int64_t)s0 + (int64_t)part * steps_per * kAccStagers;
    const int64_t u1_ = u0_ + (int64_t)steps_per * kAccStagers;
    const int32_t u0 = (int32_t)(u0_ < s1 ? u0_ : s1), u1 = (int32_t)(u1_ < s1 ? u1_ : s1);
    const int32_t nsteps = (int32_t)(((int64_t)u1 - u0 + kAccStagers - 1) / kAccStagers);
    // steps are taken kAccDepth at a time (static register rotation); the ones past the end stage row -1
    const int32_t nrounds = (nsteps + kAccDepth - 1) / kAccDepth;
    // Barrier k closes step k's staging.  The adder adds step k between barriers k and k + 1, the stagers
    // refill that buffer (step k + 2) after barrier k + 1: both sides execute nrounds * kAccDepth barriers.
    if (__builtin_amdgcn_readfirstlane(wave) != 0) {
        const int stid = tid - 64;   // staging thread (wavefronts 1..15)
        const int64_t last = (int64_t)u1 - 1;   // (nrounds > 0 implies u1 > u0)
        int32_t gr[kAccDepth];
        double gv[kAccDepth];
        auto fetch = [&](int d, int32_t step) {   // unconditional loads from clamped addresses: exact wait counts
            const int64_t j = (int64_t)u0 + (int64_t)step * kAccStagers + stid;
            // (rows16: the block-local row of a region that belongs to block b alone; 0xffff = an entry of no row)
            // (R16: the block-local row of a region that belongs to block b alone, 0xffff = an entry of no row; the staging
            // wavefronts are as busy as the adding one -- no instruction to spare for rebuilding a global row here)
            int32_t t;
            if (R16) t = (int32_t)__builtin_nontemporal_load(pr16 + (j <= last ? j : last));
            else t = __builtin_nontemporal_load(pr + (j <= last ? j : last));
            gv[d] = __builtin_nontemporal_load(px + (j <= last ? j : last));
            gr[d] = j <= last ? t : (R16 ? 0xffff : -1);
        };
        if (nrounds > 0) {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d) fetch(d, d);
        }
        for (int32_t q = 0; q < nrounds; ++q) {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d) {
                // what the adding wavefront needs and nothing more: the byte offset of the row's sum and the value.
                // Its instruction stream is the pass's critical path (one extra branch per entry there: 2.8 -> 6.6 ms),
                // so slots past the end, other blocks' entries (direct form) and invalid row indices are turned into
                // "+0.0 to a spare slot" HERE, and the adder adds unconditionally.
                const bool mine = R16 ? gr[d] != 0xffff : ((uint32_t)gr[d] < (uint32_t)nrow && (gr[d] >> shift) == b);
                st_r[(d & 1) * kAccStagers + stid] = mine ? (R16 ? gr[d] : (gr[d] & mask)) * 8 : (rows_here + (stid & 63)) * 8;   // (a spare slot per adder lane)
                st_x[(d & 1) * kAccStagers + stid] = mine ? gv[d] : 0.0;
                fetch(d, (q + 1) * kAccDepth + d);
                lds_barrier();
            }
        }
    } else {
        for (int32_t q = 0; q < nrounds; ++q) {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d) {
                lds_barrier();
                const int32_t* br = st_r + (d & 1) * kAccStagers;
                const double* bx = st_x + (d & 1) * kAccStagers;
                int32_t off[kAccStagers / 64];
                double xv[kAccStagers / 64];
#pragma unroll
                for (int u = 0; u < kAccStagers / 64; ++u) off[u] = br[u * 64 + lane];
#pragma unroll
                for (int u = 0; u < kAccStagers / 64; ++u) xv[u] = bx[u * 64 + lane];
#pragma unroll
                for (int u = 0; u < kAccStagers / 64; ++u) lds_add_f64((double*)((char*)sums + off[u]), xv[u]);
            }
        }
    }
    __syncthreads();
    const int64_t row0 = (int64_t)b << shift;
    for (int r = tid; r < rows_here; r += kAccThreads) {
        const int64_t row = row0 + r;
        if (row < nrow) {
            if (nsplit > 1) {
                part_out[(size_t)part * (size_t)nrow + (size_t)row] = sums[r];
            } else {
                double t = sums[r] + 0.0;     // a sum of -0.0 terms comes out +0.0, like the reference's accumulator
                if (MEANS) t = t / divisor;   // RcppSparse.h:153-154
                out[row] = t;
            }
        }
    }
}

// 5. (nsplit > 1) a row's sum = its parts, added in part order.
template <bool MEANS>
__global__ void rows_combine_parts_kernel(const double* __restrict__ part_out, int32_t nrow, int32_t nsplit,
                                          double* __restrict__ out, double divisor) {
#pragma clang fp contract(off)
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrow) return;
    double t = part_out[row];
    for (int q = 1; q < nsplit; ++q) t += part_out[(size_t)q * (size_t)nrow + (size_t)row];
    t = t + 0.0;
    if (MEANS) t = t / divisor;   // RcppSparse.h:153-154
    out[row] = t;
}

// 5b. the same for many parts (the direct form splits a block among up to 1024 workgroups): one wavefront per
//     row, lane l adding parts l, l + 64, ... in that order and the 64 lane sums meeting in a fixed butterfly
//     (one thread walking 256 parts costs 256 memory round trips: 0.1 ms of a 1.4 ms call)
template <bool MEANS>
__global__ __launch_bounds__(256) void rows_combine_many_parts_kernel(const double* __restrict__ part_out, int32_t nrow,
                                                                      int32_t nsplit, double* __restrict__ out,
                                                                      double divisor) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrow) return;   // (whole wavefronts)
    double t = 0.0;
    for (int q = lane; q < nsplit; q += 64) t += part_out[(size_t)q * (size_t)nrow + (size_t)row];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) t += __shfl_xor(t, d, 64);
    t = t + 0.0;
    if (MEANS) t = t / divisor;   // RcppSparse.h:153-154
    if (lane == 0) out[row] = t;
}

// 6. Several devices (or shards): a row's sum = the shards' partial sums of that row, added in SHARD order
//    (= column order, the order the reference's scatter loop meets them in), whatever way they arrived.
//    Part k is `parts + k * stride`, except part `own_idx`, which is read from `own` (a rank's own partial
//    vector is not copied into the receive area).  Two rows per thread (16-byte loads); HBM-bound:
//    8 B x nparts read + 8 B written per row.
template <bool MEANS>
__global__ __launch_bounds__(256) void rows_add_partials_kernel(const double* __restrict__ parts, int32_t nparts,
                                                                int64_t stride, const double* __restrict__ own,
                                                                int32_t own_idx, int64_t n, double* __restrict__ out,
                                                                double divisor) {
#pragma clang fp contract(off)
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (j >= n) return;
    const bool two = j + 1 < n;
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < nparts; ++k) {
        const double* src = (k == own_idx ? own : parts + (size_t)k * (size_t)stride) + j;
        const double a = src[0], b = two ? src[1] : 0.0;
        t0 = k == 0 ? a : t0 + a;
        t1 = k == 0 ? b : t1 + b;
    }
    t0 = t0 + 0.0;   // (a row whose partial sums are all -0.0 comes out +0.0, like everywhere else)
    t1 = t1 + 0.0;
    if (MEANS) {
        t0 = t0 / divisor;   // RcppSparse.h:153-154
        t1 = t1 / divisor;
    }
    out[j] = t0;
    if (two) out[j + 1] = t1;
}

hipError_t launch_add_partials(const double* parts, int32_t nparts, int64_t stride, const double* own,
                               int32_t own_idx, int64_t n, double* out, double divisor, bool means,
                               hipStream_t stream) {
    if (n <= 0 || nparts <= 0) return hipSuccess;
    const dim3 grid((unsigned)(((n + 1) / 2 + 255) / 256));
    if (means)
        hipLaunchKernelGGL(rows_add_partials_kernel<true>, grid, dim3(256), 0, stream, parts, nparts, stride, own,
                           own_idx, n, out, divisor);
    else
        hipLaunchKernelGGL(rows_add_partials_kernel<false>, grid, dim3(256), 0, stream, parts, nparts, stride, own,
                           own_idx, n, out, divisor);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// One partition pass: histogram -> scan -> first slots -> partition.  With `seg` the pass regroups one bucket
// of the two-level form (entries [seg[0], seg[1]) of src, rows [row_base, row_base + nrow_here)).
static hipError_t partition_pass(const double* src_x, const int32_t* src_i, int64_t nnz, int32_t nrow_here, int32_t shift,
                                 int32_t nblocks, const RowSumsLayout& L, int32_t* table, void* temp, double* dst_x,
                                 int32_t* dst_i, int32_t* boff, const int32_t* seg, int32_t row_base, int32_t close,
                                 hipStream_t stream, bool queue = false, bool rows16 = false) {
    uint16_t* dst_i16 = rows16 ? (uint16_t*)dst_i : nullptr;
    const size_t table_entries = (size_t)L.nsuper * (size_t)nblocks + 1;
    hipError_t e = hipMemsetAsync(table + (table_entries - 1), 0, 8, stream);   // the slot that receives the total, and the flag behind it
    if (e != hipSuccess) return e;
    auto queue_lds = [](size_t nb) { return nb * (kQueueGroup * 12 + 8 + (kPartWaves / 2) * 4) + kPartWaves * 4; };
    const size_t stage_lds = (size_t)kStageElems * 12 + ((size_t)nblocks * (2 + kPartWaves) + 1) * 4;
    const size_t part_lds = queue ? queue_lds((size_t)nblocks) : stage_lds;
    int32_t* flag = table + table_entries;   // (queue form) "the row indices are too clustered for the queues"
    static DynamicLdsLimit part_limit, queue_limit;
    e = part_limit.ensure((const void*)rows_tile_partition_kernel,
                          (int)((size_t)kStageElems * 12 + ((size_t)kPartMaxBlocks * (2 + kPartWaves) + 1) * 4));
    if (e == hipSuccess) e = queue_limit.ensure((const void*)rows_tile_partition_queue_kernel, (int)queue_lds(kQueueMaxBlocks));
    if (e != hipSuccess) return e;
    if (L.nsuper > 0) {
        hipLaunchKernelGGL(rows_tile_histogram_kernel, dim3(L.nsuper), dim3(kPartThreads), (size_t)nblocks * 4, stream,
                           src_i, nnz, nrow_here, shift, nblocks, L.super_elems, L.nsuper, table, seg, row_base,
                           queue ? kQueueGroup : 1, queue ? flag : (int32_t*)nullptr,
                           (int32_t)(((L.super_elems + kQueueTileElems - 1) / kQueueTileElems) * kQueueCellPerTile));
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    e = launch_exclusive_scan_i32(table, table, (int64_t)table_entries, 0, temp, L.temp_bytes, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rows_tile_offsets_kernel, dim3((nblocks + 1 + 255) / 256), dim3(256), 0, stream, table, nblocks,
                       L.nsuper, boff, seg, close);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (L.nsuper > 0) {
        if (queue) {   // (one-level regrouping only: no segment, rows from 0); the staged form stands by for clustered rows
            hipLaunchKernelGGL(rows_tile_partition_queue_kernel, dim3(L.nsuper), dim3(kPartThreads), part_lds, stream,
                               src_x, src_i, nnz, nrow_here, shift, nblocks, L.super_elems, L.nsuper, table, dst_x, dst_i,
                               flag, dst_i16);
            hipLaunchKernelGGL(rows_tile_partition_kernel, dim3(L.nsuper), dim3(kPartThreads), stage_lds, stream, src_x,
                               src_i, nnz, nrow_here, shift, nblocks, L.super_elems, L.nsuper, table, dst_x, dst_i, seg,
                               row_base, flag, kQueueGroup, dst_i16);
        } else {
            hipLaunchKernelGGL(rows_tile_partition_kernel, dim3(L.nsuper), dim3(kPartThreads), part_lds, stream, src_x,
                               src_i, nnz, nrow_here, shift, nblocks, L.super_elems, L.nsuper, table, dst_x, dst_i, seg,
                               row_base, (const int32_t*)nullptr, 1, dst_i16);
        }
        e = hipGetLastError();
    }
    return e;
}

// Builds the regrouped copy in `persist`; `scratch` is free again when the stream has passed this point.
hipError_t launch_row_build(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz,
                            const RowSumsLayout& L, void* persist, void* scratch, hipStream_t stream) {
    if (L.direct) return hipSuccess;   // nothing to regroup
    double* px = (double*)((char*)persist + L.vals_off);
    int32_t* pr = (int32_t*)((char*)persist + L.rows_off);
    int32_t* boff = (int32_t*)((char*)persist + L.boff_off);
    int32_t* table = (int32_t*)((char*)scratch + L.table_off);
    void* temp = (char*)scratch + L.temp_off;
    if (L.mode == 2)
        return partition_pass(d_x, d_i, nnz, nrow, L.shift + L.sub, L.ncoarse, L, table, temp, px, pr, boff, nullptr, 0, 1,
                              stream, L.aligned, L.rows16);
    // two levels: by bucket into the intermediate copy, then every bucket by its blocks into the final one
    double* mx = (double*)((char*)scratch + L.mid_vals_off);
    int32_t* mr = (int32_t*)((char*)scratch + L.mid_rows_off);
    int32_t* bucket = (int32_t*)((char*)scratch + L.bucket_off);
    hipError_t e = partition_pass(d_x, d_i, nnz, nrow, kPartBucketShift, L.nbuckets, L, table, temp, mx, mr, bucket, nullptr,
                                  0, 1, stream);
    for (int32_t s = 0; s < L.nbuckets && e == hipSuccess; ++s) {
        const int64_t row_base = (int64_t)s << kPartBucketShift;
        const int64_t rows_left = (int64_t)nrow - row_base;
        const int32_t nrow_here = (int32_t)(rows_left < ((int64_t)1 << kPartBucketShift) ? rows_left : ((int64_t)1 << kPartBucketShift));
        const int32_t blocks_here = (int32_t)(((int64_t)nrow_here + (1 << kPartShift) - 1) >> kPartShift);
        e = partition_pass(mx, mr, nnz, nrow_here, L.shift, blocks_here, L, table, temp, px, pr,
                           boff + (size_t)s * kPartTwoLevelBlocks, bucket + s, (int32_t)row_base, s == L.nbuckets - 1, stream);
    }
    return e;
}

// out[row] = the parts' sums of that row, added in part order (+ 0.0, / divisor)
static hipError_t launch_rows_combine(const double* parts, int32_t nrow, int32_t nsplit, double* d_out, double divisor,
                                      bool means, hipStream_t stream) {
    if (nsplit >= 16) {
        const dim3 wgrid((unsigned)(((int64_t)nrow + 3) / 4));
        if (means)
            hipLaunchKernelGGL(rows_combine_many_parts_kernel<true>, wgrid, dim3(256), 0, stream, parts, nrow,
                               nsplit, d_out, divisor);
        else
            hipLaunchKernelGGL(rows_combine_many_parts_kernel<false>, wgrid, dim3(256), 0, stream, parts, nrow,
                               nsplit, d_out, divisor);
        return hipGetLastError();
    }
    const dim3 cgrid((unsigned)(((int64_t)nrow + 255) / 256));
    if (means)
        hipLaunchKernelGGL(rows_combine_parts_kernel<true>, cgrid, dim3(256), 0, stream, parts, nrow, nsplit,
                           d_out, divisor);
    else
        hipLaunchKernelGGL(rows_combine_parts_kernel<false>, cgrid, dim3(256), 0, stream, parts, nrow, nsplit,
                           d_out, divisor);
    return hipGetLastError();
}

// Row sums / means from the regrouped copy in `persist` (direct form: from the caller's x / i).
hipError_t launch_row_reduce(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz,
                             const RowSumsLayout& L, void* persist, double* d_out,
                             double divisor, bool means, const LaunchPlan& colsums_plan, hipStream_t stream) {
    (void)colsums_plan;
    if (nrow <= 0) return hipSuccess;
    const bool direct = L.direct;
    const double* px = direct ? d_x : (const double*)((char*)persist + L.vals_off);
    const int32_t* pr = direct ? d_i : (const int32_t*)((char*)persist + L.rows_off);
    const int32_t* boff = (const int32_t*)((char*)persist + L.boff_off);
    const uint16_t* pr16 = (!direct && L.rows16) ? (const uint16_t*)pr : nullptr;
    const size_t acc_lds = ((size_t)8 << L.shift) + 64 * 8 + (size_t)2 * kAccStagers * 12;
    static DynamicLdsLimit acc_limit_means, acc_limit_sums, acc_limit_means16, acc_limit_sums16;
    hipError_t e = acc_limit_means.ensure((const void*)rows_tile_accumulate_kernel<true>, (int)acc_lds);
    if (e == hipSuccess) e = acc_limit_sums.ensure((const void*)rows_tile_accumulate_kernel<false>, (int)acc_lds);
    if (e == hipSuccess) e = acc_limit_means16.ensure((const void*)rows_tile_accumulate_kernel<true, true>, (int)acc_lds);
    if (e == hipSuccess) e = acc_limit_sums16.ensure((const void*)rows_tile_accumulate_kernel<false, true>, (int)acc_lds);
    if (e != hipSuccess) return e;
    double* parts = (double*)((char*)persist + L.partial_off);
    const dim3 grid((unsigned)L.nblocks * (unsigned)L.nsplit);
    if (pr16 && means)
        hipLaunchKernelGGL((rows_tile_accumulate_kernel<true, true>), grid, dim3(kAccThreads), acc_lds, stream, px, pr,
                           boff, (int64_t)-1, nrow, L.shift, L.sub, L.nsplit, d_out, parts, divisor, pr16);
    else if (pr16)
        hipLaunchKernelGGL((rows_tile_accumulate_kernel<false, true>), grid, dim3(kAccThreads), acc_lds, stream, px, pr,
                           boff, (int64_t)-1, nrow, L.shift, L.sub, L.nsplit, d_out, parts, divisor, pr16);
    else if (means)
        hipLaunchKernelGGL(rows_tile_accumulate_kernel<true>, grid, dim3(kAccThreads), acc_lds, stream, px, pr,
                           boff, direct ? nnz : (int64_t)-1, nrow, L.shift, L.sub, L.nsplit, d_out, parts, divisor, pr16);
    else
        hipLaunchKernelGGL(rows_tile_accumulate_kernel<false>, grid, dim3(kAccThreads), acc_lds, stream, px, pr,
                           boff, direct ? nnz : (int64_t)-1, nrow, L.shift, L.sub, L.nsplit, d_out, parts, divisor, pr16);
    e = hipGetLastError();
    if (e != hipSuccess || L.nsplit <= 1) return e;
    return launch_rows_combine(parts, nrow, L.nsplit, d_out, divisor, means, stream);
}

// ---------------------------------------------------------------------------------------------
// segments form (round 3; behind a handle, which holds p[]): no regrouping when the columns are long
// ---------------------------------------------------------------------------------------------
// Inside a column of a dgCMatrix the rows ascend, so the entries of column c that belong to row block b are ONE
// contiguous piece [T[b][c], T[b + 1][c]) of x / i.  With that table (a binary search per (block, column), built once
// per handle; T[0] = p[c], T[nblocks] = p[c + 1]) the accumulate pass of every block reads exactly its own entries from
// the caller's arrays: 12 B/nnz in total, where the direct form reads all entries once per block (2-4 blocks: 24-48
// B/nnz) and the partition forms move ~40 B/nnz and keep a 12 B/nnz copy.  It pays when a (column, block) piece has
// a hundred entries or more -- matrices of long columns over 16385 .. ~1e6 rows -- and needs rows that really ascend:
// the handle checks that once on the device (rows_sorted_check_kernel) and keeps the other forms for matrices that fail.
//   Work: `nsplit` workgroups per row block, each over a range of whole columns (cuts balanced by entries).  The 15
//   staging wavefronts take the range's columns round-robin and walk their column's piece 64 entries per step; the
//   sixteenth wavefront adds, exactly as in rows_tile_accumulate_kernel.  A row's terms are added in a fixed order
//   (step, slot): bit-stable; within the usual 1e-12 * sum|x| of the reference's order.
constexpr int kSegMinPiece = 128;         // mean entries per (column, block) piece from which the form is chosen
constexpr int kSegMinColumnsPerPart = 30; // a workgroup's 15 staging wavefronts want two columns each at least

__global__ __launch_bounds__(256) void rows_sorted_check_kernel(const int32_t* __restrict__ ri,
                                                                const int32_t* __restrict__ p, int32_t ncol,
                                                                int64_t nnz, int32_t* __restrict__ flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (e >= nnz) return;
    if (ri[e] >= ri[e - 1]) return;
    // a descent: fine only where a column starts (p[c] == e for the column c that holds entry e)
    int32_t lo = 0, hi = ncol;   // invariant: p[lo] <= e < p[hi]
    while (hi - lo > 1) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)p[mid] <= e) lo = mid; else hi = mid;
    }
    if ((int64_t)p[lo] != e) *flag = 1;
}

// T[b][c] for b = 0 .. nblocks: first entry of column c whose row is >= b << shift (T[0] = p[c], T[nblocks] = p[c + 1])
__global__ __launch_bounds__(256) void rows_segment_table_kernel(const int32_t* __restrict__ ri,
                                                                 const int32_t* __restrict__ p, int32_t ncol,
                                                                 int32_t nblocks, int32_t shift,
                                                                 int32_t* __restrict__ table) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)(nblocks + 1) * ncol) return;
    const int32_t b = (int32_t)(t / ncol), c = (int32_t)(t - (int64_t)b * ncol);
    int32_t lo = p[c], hi = p[c + 1];
    if (b == 0) {
        table[t] = lo;
    } else if (b == nblocks) {
        table[t] = hi;
    } else {
        const int64_t first_row = (int64_t)b << shift;
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)ri[mid] < first_row) lo = mid + 1; else hi = mid;
        }
        table[t] = lo;
    }
}

// cuts[k] = first column whose entries start at or after k * nnz / nsplit (cuts[0] = 0, cuts[nsplit] = ncol)
__global__ void rows_column_cuts_kernel(const int32_t* __restrict__ p, int32_t ncol, int64_t nnz, int32_t nsplit,
                                        int32_t* __restrict__ cuts) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nsplit) return;
    if (k == 0 || k == nsplit) {
        cuts[k] = k == 0 ? 0 : ncol;
        return;
    }
    const int64_t target = nnz / nsplit * k + nnz % nsplit * k / nsplit;
    int32_t lo = 0, hi = ncol;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)p[mid] < target) lo = mid + 1; else hi = mid;
    }
    cuts[k] = lo;
}

template <bool MEANS>
__global__ __launch_bounds__(kAccThreads) void rows_segments_accumulate_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, const int32_t* __restrict__ table,
    const int32_t* __restrict__ cuts, int32_t ncol, int32_t nrow, int32_t shift, int32_t nsplit,
    double* __restrict__ out, double* __restrict__ part_out, double divisor) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    double* sums = (double*)s_raw;                          // 1 << shift (+ 64 spare slots: where nothing-to-add goes)
    double* st_x = sums + ((size_t)1 << shift) + 64;        // 2 x kAccStagers
    int32_t* st_r = (int32_t*)(st_x + 2 * kAccStagers);     // 2 x kAccStagers: byte offsets into sums
    __shared__ int32_t s_steps[kAccThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / nsplit, part = blockIdx.x - b * nsplit;
    const int rows_here = 1 << shift, mask = rows_here - 1;
    for (int r = tid; r < rows_here; r += kAccThreads) sums[r] = 0.0;
    const int32_t ca = cuts[part], cb = cuts[part + 1];
    const int32_t* tb = table + (size_t)b * ncol;        // piece starts of this block ...
    const int32_t* te = tb + ncol;                       // ... and ends (= the next block's starts)
    constexpr int kStagerWaves = kAccStagers / 64;
    // steps this staging wavefront needs: its columns' pieces, 64 entries per step
    if (wave != 0) {
        int32_t cnt = 0;
        for (int64_t c = (int64_t)ca + (wave - 1) + (int64_t)kStagerWaves * lane; c < cb; c += (int64_t)kStagerWaves * 64) {
            const int32_t len = te[c] - tb[c];
            cnt += len > 0 ? (len + 63) >> 6 : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if (lane == 0) s_steps[wave] = cnt;
    } else if (lane == 0) {
        s_steps[0] = 0;
    }
    __syncthreads();
    int32_t nsteps = 0;
#pragma unroll
    for (int w = 0; w < kAccThreads / 64; ++w) nsteps = s_steps[w] > nsteps ? s_steps[w] : nsteps;
    const int32_t nrounds = (nsteps + kAccDepth - 1) / kAccDepth;
    if (wave != 0) {
        const int stid = tid - 64;
        // this wavefront's place in its column list: the piece [pos, end) of column c; the next column's piece is
        // requested when this one is entered (all wave-uniform: scalar loads)
        int32_t c = ca + (wave - 1);
        int32_t pos = 0, end = 0, npos = 0, nend = 0;
        if (c < cb) {
            pos = tb[c];
            end = te[c];
        }
        if (c + kStagerWaves < cb) {
            npos = tb[c + kStagerWaves];
            nend = te[c + kStagerWaves];
        }
        int32_t gr[kAccDepth];
        double gv[kAccDepth];
        auto fetch = [&](int d) {
            while (pos >= end && c < cb) {   // (empty pieces, and the end of a piece: on to the wavefront's next column)
                c += kStagerWaves;
                pos = npos;
                end = nend;
                npos = nend = 0;
                if (c + kStagerWaves < cb) {
                    npos = tb[c + kStagerWaves];
                    nend = te[c + kStagerWaves];
                }
                if (c >= cb) pos = end = 0;
            }
            const int32_t n = end - pos < 64 ? end - pos : 64;   // (0 once the columns are used up)
            // unconditional loads from clamped addresses (exact wait counts); lanes past the piece stage "nothing"
            const int32_t j = lane < n ? (int32_t)((uint32_t)pos + (uint32_t)lane) : (n > 0 ? pos : 0);
            const int32_t t = __builtin_nontemporal_load(ri + j);
            gv[d] = __builtin_nontemporal_load(x + j);
            gr[d] = lane < n ? t : -1;
            pos += n;
        };
        if (nrounds > 0) {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d) fetch(d);
        }
        for (int32_t q = 0; q < nrounds; ++q) {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d) {
                const bool mine = (uint32_t)gr[d] < (uint32_t)nrow && (gr[d] >> shift) == b;
                st_r[(d & 1) * kAccStagers + stid] = mine ? (gr[d] & mask) * 8 : (rows_here + (stid & 63)) * 8;
                st_x[(d & 1) * kAccStagers + stid] = mine ? gv[d] : 0.0;
                fetch(d);
                lds_barrier();
            }
        }
    } else {
        for (int32_t q = 0; q < nrounds; ++q) {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d) {
                lds_barrier();
                const int32_t* br = st_r + (d & 1) * kAccStagers;
                const double* bx = st_x + (d & 1) * kAccStagers;
                int32_t off[kAccStagers / 64];
                double xv[kAccStagers / 64];
#pragma unroll
                for (int u = 0; u < kAccStagers / 64; ++u) off[u] = br[u * 64 + lane];
#pragma unroll
                for (int u = 0; u < kAccStagers / 64; ++u) xv[u] = bx[u * 64 + lane];
#pragma unroll
                for (int u = 0; u < kAccStagers / 64; ++u) lds_add_f64((double*)((char*)sums + off[u]), xv[u]);
            }
        }
    }
    __syncthreads();
    const int64_t row0 = (int64_t)b << shift;
    for (int r = tid; r < rows_here; r += kAccThreads) {
        const int64_t row = row0 + r;
        if (row < nrow) {
            if (nsplit > 1) {
                part_out[(size_t)part * (size_t)nrow + (size_t)row] = sums[r];
            } else {
                double t = sums[r] + 0.0;
                if (MEANS) t = t / divisor;
                out[row] = t;
            }
        }
    }
}

bool row_segments_applicable(int32_t nrow, int32_t ncol, int64_t nnz, bool force) {
    const int64_t nblocks = ((int64_t)nrow + (1 << kPartShift) - 1) >> kPartShift;
    if (nblocks < 2 || ncol < 1 || nnz < 1 || nnz > 0x7fffffffll) return false;
    if ((nblocks + 1) * (int64_t)ncol > 0x7fffffffll) return false;   // (table entries)
    if (force) return true;
    return ncol >= kSegMinColumnsPerPart && nnz / ((int64_t)ncol * nblocks) >= kSegMinPiece;
}

hipError_t plan_row_segments(int32_t nrow, int32_t ncol, int64_t nnz, RowSegmentsLayout* L) {
    memset(L, 0, sizeof(*L));
    L->shift = kPartShift;
    L->nblocks = (int32_t)(((int64_t)nrow + (1 << kPartShift) - 1) >> kPartShift);
    int ns = accumulate_split(L->nblocks, nnz / L->nblocks, 1 << kPartShift);
    const int by_columns = ncol / kSegMinColumnsPerPart;   // parts of at least 30 columns
    if (ns > by_columns) ns = by_columns;
    L->nsplit = ns < 1 ? 1 : ns;
    size_t off = 0;
    L->table_off = off;   off = align_up(off + (size_t)(L->nblocks + 1) * (size_t)ncol * 4, 256);
    L->cuts_off = off;    off = align_up(off + ((size_t)L->nsplit + 1) * 4, 256);
    L->flag_off = off;    off = align_up(off + 4, 256);
    L->partial_off = off;
    if (L->nsplit > 1) off = align_up(off + (size_t)L->nsplit * (size_t)nrow * 8, 256);
    L->bytes = off;
    return hipSuccess;
}

hipError_t launch_rows_sorted_check(const int32_t* d_i, const int32_t* d_p, int32_t ncol, int64_t nnz,
                                    int32_t* d_flag, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_flag, 0, 4, stream);
    if (e != hipSuccess || nnz < 2) return e;
    hipLaunchKernelGGL(rows_sorted_check_kernel, dim3((unsigned)((nnz - 1 + 255) / 256)), dim3(256), 0, stream, d_i,
                       d_p, ncol, nnz, d_flag);
    return hipGetLastError();
}

hipError_t launch_row_segments_build(const int32_t* d_i, const int32_t* d_p, int32_t ncol, int64_t nnz,
                                     const RowSegmentsLayout& L, void* persist, hipStream_t stream) {
    int32_t* table = (int32_t*)((char*)persist + L.table_off);
    int32_t* cuts = (int32_t*)((char*)persist + L.cuts_off);
    const int64_t cells = (int64_t)(L.nblocks + 1) * ncol;
    hipLaunchKernelGGL(rows_segment_table_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, d_i,
                       d_p, ncol, L.nblocks, L.shift, table);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rows_column_cuts_kernel, dim3((unsigned)((L.nsplit + 1 + 255) / 256)), dim3(256), 0, stream,
                       d_p, ncol, nnz, L.nsplit, cuts);
    return hipGetLastError();
}

hipError_t launch_row_segments_reduce(const double* d_x, const int32_t* d_i, int32_t nrow, int32_t ncol,
                                      const RowSegmentsLayout& L, void* persist, double* d_out, double divisor,
                                      bool means, hipStream_t stream) {
    if (nrow <= 0) return hipSuccess;
    const int32_t* table = (const int32_t*)((char*)persist + L.table_off);
    const int32_t* cuts = (const int32_t*)((char*)persist + L.cuts_off);
    double* parts = (double*)((char*)persist + L.partial_off);
    const size_t acc_lds = ((size_t)8 << L.shift) + 64 * 8 + (size_t)2 * kAccStagers * 12;
    static DynamicLdsLimit lim_means, lim_sums;
    hipError_t e = lim_means.ensure((const void*)rows_segments_accumulate_kernel<true>, (int)acc_lds);
    if (e == hipSuccess) e = lim_sums.ensure((const void*)rows_segments_accumulate_kernel<false>, (int)acc_lds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)L.nblocks * (unsigned)L.nsplit);
    if (means)
        hipLaunchKernelGGL(rows_segments_accumulate_kernel<true>, grid, dim3(kAccThreads), acc_lds, stream, d_x, d_i,
                           table, cuts, ncol, nrow, L.shift, L.nsplit, d_out, parts, divisor);
    else
        hipLaunchKernelGGL(rows_segments_accumulate_kernel<false>, grid, dim3(kAccThreads), acc_lds, stream, d_x, d_i,
                           table, cuts, ncol, nrow, L.shift, L.nsplit, d_out, parts, divisor);
    e = hipGetLastError();
    if (e != hipSuccess || L.nsplit <= 1) return e;
    return launch_rows_combine(parts, nrow, L.nsplit, d_out, divisor, means, stream);
}

}  // namespace rsp

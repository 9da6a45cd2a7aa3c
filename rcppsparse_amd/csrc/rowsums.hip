// rowsums.hip -- Matrix::rowSums / rowMeans on the device ("next" row f1 of SURVEY.md 8f).
//
// Reference inst/include/RcppSparse.h:138-144 scatters sums(i[j]) += x[j] while walking the
// columns.  On the device that is a reduction by key with 1e7 keys and no locality in the key.
// Three forms, chosen in plan_row_sums:
//
//   tile partition (one-shot calls, rsp_row_sums_device, up to 1.36e7 rows) -- all hand-written:
//     the entries are regrouped by ROW BLOCK (16384 rows: what one CU's LDS holds as sums) in ONE
//     pass, then added up block by block.
//       1. rows_tile_histogram_kernel   entries per (block, supertile)              reads  4 B/nnz
//       2. one exclusive scan over that table (rocPRIM; a few MB) = first output slot of every pair
//       3. rows_tile_partition_kernel   a workgroup walks its supertile in tiles of 8192 entries:
//          ranks them per block with wave-private LDS counters, SORTS THE TILE BY BLOCK IN LDS, and
//          writes every block's run to its cursor -- consecutive lanes store consecutive slots, so
//          the ~600 output streams leave as runs of ~13 entries instead of single 8-byte stores
//                                                                reads 12 B/nnz, writes 12 B/nnz
//       4. rows_tile_accumulate_kernel  one 16-wave workgroup per block: the block's sums live in
//          128 KB of LDS, tiles of entries are staged once and every wave adds the rows it owns
//          (row mod 16) with ds_add_f64, in slot order              reads 12 B/nnz, writes 8 B/row
//     About 40 B/nnz of traffic and a workspace of 12 B/nnz + the count table.  An entry's slot is a
//     function of the data alone (no global atomics, wave-private counters combined in a fixed
//     order), and a row's terms are added by one wavefront in slot order: bit-stable run to run.
//
//   block sort (one-shot calls on matrices with more rows): a stable rocPRIM radix sort over the
//     block bits of the row index only (4096-row blocks), then rows_block_accumulate_kernel (one
//     wavefront per block, sums in 32 KB of LDS).
//
//   row form (the handle API, which keeps it for repeated calls): full stable sort by row,
//     row offsets by a vectorised lower_bound, then the column-sum kernels on the row-major
//     values (8 B/nnz per repeated call instead of 12).
//
// All are deterministic and within the usual 1e-12 * sum|x| of the reference's order.
// Round-2 history (profiles/r02_rowsums.md): partitioning WITHOUT sorting each tile in LDS first
// (every lane storing its entry straight to its block's cursor) took 47 ms for the scatter alone.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kRowBlockShift = 12;   // block sort: 4096 rows per block, 32 KB of LDS sums per wavefront, 5 per CU

// tile partition
constexpr int kPartShift = 14;            // 16384 rows per block: 128 KB of LDS sums per workgroup
constexpr int kPartMaxBlocks = 832;       // LDS counters / cursors of the partition kernel (with the tile: < 160 KB)
constexpr int kPartThreads = 1024;        // partition workgroup: 16 wavefronts
constexpr int kPartWaves = kPartThreads / 64;
constexpr int kPartPerThread = 8;
constexpr int kTileElems = kPartThreads * kPartPerThread;   // 8192 entries sorted in LDS at a time
constexpr int kTilesPerSuper = 20;        // a workgroup's supertile: 163 840 entries
constexpr size_t kCountTableMaxBytes = 64u << 20;
constexpr int kAccThreads = 1024;         // accumulate workgroup: 16 wavefronts stage the entries, ONE adds them
constexpr int kAccStage = 2048;           // entries staged per step

static unsigned key_bits(int32_t nrow) {
    unsigned b = 1;
    while (b < 31 && (1u << b) < (unsigned)nrow) ++b;
    return b;
}

// first bit the block sort looks at (all bits of a matrix with a single block: nothing to sort by)
static unsigned block_begin_bit(int32_t nrow) {
    const unsigned hi = key_bits(nrow);
    return hi > (unsigned)kRowBlockShift ? (unsigned)kRowBlockShift : hi;
}

// block id of an entry, as the offsets search sees the sorted row indices
struct RowBlockOf {
    int shift;
    __host__ __device__ uint32_t operator()(uint32_t row) const { return row >> shift; }
};

// ---------------------------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------------------------
hipError_t plan_row_sums(int32_t nrow, int64_t nnz, size_t colsums_ws_bytes, bool keep_row_form,
                         RowSumsLayout* L) {
    memset(L, 0, sizeof(*L));
    size_t sort_bytes = 0, search_bytes = 0, off = 0;
    hipError_t e;
    const int64_t part_blocks = ((int64_t)nrow + (1 << kPartShift) - 1) >> kPartShift;
    if (!keep_row_form && part_blocks <= kPartMaxBlocks) {
        L->mode = 2;
        L->shift = kPartShift;
        L->nblocks = (int32_t)(part_blocks > 0 ? part_blocks : 1);
        int64_t super = (int64_t)kTileElems * kTilesPerSuper;   // grow until the count table fits its budget
        while (((nnz + super - 1) / super) * L->nblocks * 4 > (int64_t)kCountTableMaxBytes) super *= 2;
        L->super_elems = super;
        L->nsuper = (int32_t)((nnz + super - 1) / super);
        const size_t table_entries = (size_t)L->nsuper * (size_t)L->nblocks + 1;   // + the total
        size_t temp = 0;
        e = rocprim::exclusive_scan(nullptr, temp, (const int32_t*)nullptr, (int32_t*)nullptr, 0, table_entries,
                                    rocprim::plus<int32_t>(), (hipStream_t)0);
        if (e != hipSuccess) return e;
        L->vals_off = off;  off = align_up(off + (size_t)nnz * 8, 256);                // x grouped by row block
        L->rows_off = off;  off = align_up(off + (size_t)nnz * 4, 256);                // their row indices
        L->boff_off = off;  off = align_up(off + ((size_t)L->nblocks + 1) * 4, 256);   // first slot of every block
        L->persistent_bytes = off;
        off = 0;
        L->table_off = off; off = align_up(off + table_entries * 4, 256);
        L->temp_off = off;
        L->temp_bytes = temp;
        L->scratch_bytes = align_up(off + temp, 256);
        return hipSuccess;
    }
    if (!keep_row_form) {
        L->mode = 0;
        L->shift = kRowBlockShift;
        L->nblocks = (int32_t)(((int64_t)nrow + (1 << kRowBlockShift) - 1) >> kRowBlockShift);
        if (L->nblocks < 1) L->nblocks = 1;
        if (block_begin_bit(nrow) < key_bits(nrow)) {   // (a matrix of one block is copied, not sorted)
            e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                          (const double*)nullptr, (double*)nullptr, (size_t)nnz,
                                          block_begin_bit(nrow), key_bits(nrow), (hipStream_t)0);
            if (e != hipSuccess) return e;
        }
        auto ids = rocprim::make_transform_iterator((const uint32_t*)nullptr, RowBlockOf{kRowBlockShift});
        e = rocprim::lower_bound(nullptr, search_bytes, ids, rocprim::counting_iterator<uint32_t>(0),
                                 (int32_t*)nullptr, (size_t)nnz, (size_t)L->nblocks + 1, rocprim::less<uint32_t>(),
                                 (hipStream_t)0);
        if (e != hipSuccess) return e;
        L->vals_off = off;  off = align_up(off + (size_t)nnz * 8, 256);                // x grouped by row block
        L->rows_off = off;  off = align_up(off + (size_t)nnz * 4, 256);                // their row indices
        L->boff_off = off;  off = align_up(off + ((size_t)L->nblocks + 1) * 4, 256);   // first entry of every block
        L->persistent_bytes = off;
        off = 0;
    } else {
        L->mode = 1;
        e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (const double*)nullptr, (double*)nullptr, (size_t)nnz, 0u, key_bits(nrow),
                                      (hipStream_t)0);
        if (e != hipSuccess) return e;
        e = rocprim::lower_bound(nullptr, search_bytes, (const uint32_t*)nullptr,
                                 rocprim::counting_iterator<uint32_t>(0), (int32_t*)nullptr, (size_t)nnz,
                                 (size_t)nrow + 1, rocprim::less<uint32_t>(), (hipStream_t)0);
        if (e != hipSuccess) return e;
        L->vals_off = off;  off = align_up(off + (size_t)nnz * 8, 256);         // x sorted by row
        L->prow_off = off;  off = align_up(off + ((size_t)nrow + 1) * 4, 256);  // row offsets
        L->colsums_off = off; off = align_up(off + colsums_ws_bytes, 256);      // chunk carries
        L->persistent_bytes = off;
        off = 0;
        L->keys_off = off;  off = align_up(off + (size_t)nnz * 4, 256);         // sorted row indices
    }
    L->temp_bytes = sort_bytes > search_bytes ? sort_bytes : search_bytes;
    L->temp_off = off;
    L->scratch_bytes = align_up(off + L->temp_bytes, 256);
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// block form: accumulate
// ---------------------------------------------------------------------------------------------
// sums[r] += v as one LDS instruction (an IEEE double add performed by the LDS unit).  A wave's LDS
// instructions execute in issue order; lanes of one instruction that hit the same row are serialised
// by the hardware, always the same way.
__device__ __forceinline__ void lds_add_f64(double* a, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)a, v);
}

// One wavefront per row block: its entries (storage order) into LDS sums, then out.
template <bool MEANS>
__global__ __launch_bounds__(64) void rows_block_accumulate_kernel(
    const double* __restrict__ px, const int32_t* __restrict__ pr, const int32_t* __restrict__ boff,
    int32_t nrow, int32_t shift, double* __restrict__ out, double divisor) {
#pragma clang fp contract(off)
    extern __shared__ double s_sums[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int rows_here = 1 << shift;
    const int mask = rows_here - 1;
    for (int r = lane; r < rows_here; r += 64) s_sums[r] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int32_t s0 = boff[b], s1 = boff[b + 1];
    // 16 steps of 64 entries in flight (12 KB per wavefront, 5 wavefronts per CU)
    for (int32_t s = s0; s < s1; s += 16 * 64) {
        int32_t r[16];
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int32_t j = s + k * 64 + lane;
            const bool in = j < s1;
            r[k] = in ? pr[j] : -1;
            v[k] = in ? px[j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (r[k] >= 0) lds_add_f64(&s_sums[r[k] & mask], v[k]);
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t row0 = (int64_t)b << shift;
    for (int r = lane; r < rows_here; r += 64) {
        const int64_t row = row0 + r;
        if (row < nrow) {
            double t = s_sums[r] + 0.0;   // a sum of -0.0 terms comes out +0.0, like the reference's accumulator
            if (MEANS) t = t / divisor;   // RcppSparse.h:153-154
            out[row] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tile partition: histogram, partition, accumulate
// ---------------------------------------------------------------------------------------------
// 1. entries per (row block, supertile).  The table is [block][supertile], so that ONE flat exclusive
//    scan yields for every pair the number of entries in earlier blocks plus those of the same block
//    in earlier supertiles: its first output slot.  (Row indices outside [0, nrow) -- not a valid
//    dgCMatrix -- are neither counted here nor moved below.)
__global__ __launch_bounds__(kPartThreads) void rows_tile_histogram_kernel(
    const int32_t* __restrict__ ri, int64_t nnz, int32_t nrow, int32_t shift, int32_t nblocks,
    int64_t super_elems, int32_t nsuper, int32_t* __restrict__ table) {
    extern __shared__ int32_t s_hist[];
    const int tid = threadIdx.x;
    const int s = blockIdx.x;
    for (int b = tid; b < nblocks; b += kPartThreads) s_hist[b] = 0;
    __syncthreads();
    const int64_t e0 = (int64_t)s * super_elems;
    const int64_t e1 = e0 + super_elems < nnz ? e0 + super_elems : nnz;
    for (int64_t e = e0; e < e1; e += 8 * kPartThreads) {   // 8 coalesced loads in flight per thread
        int32_t r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t j = e + k * kPartThreads + tid;
            r[k] = j < e1 ? ri[j] : -1;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((uint32_t)r[k] < (uint32_t)nrow) atomicAdd(&s_hist[(uint32_t)r[k] >> shift], 1);
    }
    __syncthreads();
    for (int b = tid; b < nblocks; b += kPartThreads) table[(size_t)b * nsuper + s] = s_hist[b];
}

// 3. the partition pass.  LDS: the tile sorted by block (values, row indices), one cursor per block
//    (next output slot of this supertile), the tile's first position per block, and one counter per
//    (wavefront, block).
__global__ __launch_bounds__(kPartThreads) void rows_tile_partition_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, int64_t nnz, int32_t nrow, int32_t shift,
    int32_t nblocks, int64_t super_elems, int32_t nsuper, const int32_t* __restrict__ first_slot,
    double* __restrict__ px, int32_t* __restrict__ pr) {
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    double* stage_x = (double*)s_raw;                                   // kTileElems
    int32_t* stage_r = (int32_t*)(stage_x + kTileElems);                // kTileElems
    int32_t* cursor = stage_r + kTileElems;                             // nblocks
    int32_t* tstart = cursor + nblocks;                                 // nblocks + 1
    int32_t* cnt = tstart + nblocks + 1;                                // kPartWaves x nblocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    for (int b = tid; b < nblocks; b += kPartThreads) cursor[b] = first_slot[(size_t)b * nsuper + s];
    const int64_t e0 = (int64_t)s * super_elems;
    const int64_t e1 = e0 + super_elems < nnz ? e0 + super_elems : nnz;
    int32_t* mycnt = cnt + (size_t)wave * nblocks;
    // this thread's entries of the current tile (coalesced across the workgroup); the next tile's are
    // fetched as soon as the current ones sit in LDS, so their latency hides behind the write-out
    int32_t r[kPartPerThread], rank[kPartPerThread];
    double v[kPartPerThread];
    auto fetch = [&](int64_t tile) {
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k) {
            const int64_t j = tile + k * kPartThreads + tid;
            const bool in = j < e1;
            r[k] = in ? ri[j] : -1;
            v[k] = in ? x[j] : 0.0;
        }
    };
    fetch(e0);
    for (int64_t tile = e0; tile < e1; tile += kTileElems) {
        for (int k = tid; k < kPartWaves * nblocks; k += kPartThreads) cnt[k] = 0;
        __syncthreads();   // (also: cursors initialised / updated, previous tile written out)
        // rank of every entry among the entries of the same block handled by the same wavefront:
        // program order inside the wavefront, hardware order inside one LDS instruction -- both fixed
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k)
            rank[k] = (uint32_t)r[k] < (uint32_t)nrow ? atomicAdd(&mycnt[(uint32_t)r[k] >> shift], 1) : -1;
        __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
        // the tile, sorted by block, into LDS
#pragma unroll
        for (int k = 0; k < kPartPerThread; ++k)
            if (rank[k] >= 0) {
                const int b = (uint32_t)r[k] >> shift;
                const int pos = tstart[b] + mycnt[b] + rank[k];
                stage_x[pos] = v[k];
                stage_r[pos] = r[k];
            }
        fetch(tile + kTileElems);   // (past the supertile: nothing is loaded)
        __syncthreads();
        // ... and out: consecutive threads, consecutive positions, consecutive slots inside a block's run
        const int total = tstart[nblocks];
        for (int j = tid; j < total; j += kPartThreads) {
            const int32_t rr = stage_r[j];
            const int b = (uint32_t)rr >> shift;
            const int32_t dest = cursor[b] + (j - tstart[b]);
            px[dest] = stage_x[j];
            pr[dest] = rr;
        }
        __syncthreads();
        for (int b = tid; b < nblocks; b += kPartThreads) cursor[b] += tstart[b + 1] - tstart[b];
    }
}

// boff[b] = first slot of block b (= the scanned table's entry for supertile 0), boff[nblocks] = number of
// entries with a valid row index (the scan's last element)
__global__ void rows_tile_offsets_kernel(const int32_t* __restrict__ first_slot, int32_t nblocks, int32_t nsuper,
                                         int32_t* __restrict__ boff) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) boff[b] = first_slot[(size_t)b * nsuper];
    if (b == nblocks) boff[b] = first_slot[(size_t)nblocks * nsuper];
}

// 4. one 16-wave workgroup per row block.  All wavefronts stage the block's entries (coalesced loads,
//    the next step's already in flight); ONE wavefront adds them, 64 per LDS instruction, in slot order.
//    (An LDS double add costs the same ~16 cycles whether 4 or 64 of its lanes are active, so letting
//    every wavefront pick "its" rows out of each step is 16 times the LDS work: 9.5 ms instead of 2.)
template <bool MEANS>
__global__ __launch_bounds__(kAccThreads) void rows_tile_accumulate_kernel(
    const double* __restrict__ px, const int32_t* __restrict__ pr, const int32_t* __restrict__ boff,
    int32_t nrow, int32_t shift, double* __restrict__ out, double divisor) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    double* sums = (double*)s_raw;                          // 1 << shift
    double* st_x = sums + ((size_t)1 << shift);             // kAccStage
    int32_t* st_r = (int32_t*)(st_x + kAccStage);           // kAccStage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int rows_here = 1 << shift, mask = rows_here - 1;
    for (int r = tid; r < rows_here; r += kAccThreads) sums[r] = 0.0;
    const int32_t s0 = boff[b], s1 = boff[b + 1];
    // the entries of the next TWO steps travel in registers (two sets, refilled right after they have been
    // staged) while the current step is added: 48 KB of loads in flight per CU
    struct Regs {
        int32_t r0, r1;
        double v0, v1;
    };
    auto fetch = [&](Regs& g, int32_t t) {
        const int32_t j0 = t + tid, j1 = t + kAccThreads + tid;
        g.r0 = j0 < s1 ? pr[j0] : -1;
        g.v0 = j0 < s1 ? px[j0] : 0.0;
        g.r1 = j1 < s1 ? pr[j1] : -1;
        g.v1 = j1 < s1 ? px[j1] : 0.0;
    };
    auto step = [&](Regs& g, int32_t refill_from) {
        __syncthreads();   // the previous step has been added (first pass: the sums are zeroed)
        st_r[tid] = g.r0;
        st_x[tid] = g.v0;
        st_r[tid + kAccThreads] = g.r1;
        st_x[tid + kAccThreads] = g.v1;
        __syncthreads();
        fetch(g, refill_from);
        if (wave == 0) {   // (slots past the end of the block hold row -1)
#pragma unroll
            for (int q0 = 0; q0 < kAccStage; q0 += 8 * 64) {
                int32_t rr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) rr[u] = st_r[q0 + u * 64 + lane];
                double xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = st_x[q0 + u * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (rr[u] >= 0) lds_add_f64(&sums[rr[u] & mask], xv[u]);
            }
        }
    };
    Regs ga, gb;
    fetch(ga, s0);
    fetch(gb, s0 + kAccStage);
    for (int32_t t = s0; t < s1; t += 2 * kAccStage) {
        step(ga, t + 2 * kAccStage);
        if (t + kAccStage < s1) step(gb, t + 3 * kAccStage);   // (uniform: every thread takes the same path)
    }
    __syncthreads();
    const int64_t row0 = (int64_t)b << shift;
    for (int r = tid; r < rows_here; r += kAccThreads) {
        const int64_t row = row0 + r;
        if (row < nrow) {
            double t = sums[r] + 0.0;     // a sum of -0.0 terms comes out +0.0, like the reference's accumulator
            if (MEANS) t = t / divisor;   // RcppSparse.h:153-154
            out[row] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// Builds the row-wise form in `persist`; `scratch` is free again when the stream has passed this point.
hipError_t launch_row_build(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz,
                            const RowSumsLayout& L, void* persist, void* scratch, hipStream_t stream) {
    void* temp = (char*)scratch + L.temp_off;
    size_t temp_bytes = L.temp_bytes;
    hipError_t e = hipSuccess;
    if (L.mode == 1) {
        double* vals = (double*)((char*)persist + L.vals_off);
        int32_t* prow = (int32_t*)((char*)persist + L.prow_off);
        uint32_t* keys = (uint32_t*)((char*)scratch + L.keys_off);
        if (nnz > 0) {
            e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)d_i, keys, d_x, vals, (size_t)nnz, 0u,
                                          key_bits(nrow), stream);
            if (e != hipSuccess) return e;
        }
        temp_bytes = L.temp_bytes;
        // prow[r] = first position whose row index is >= r  (r = 0..nrow; prow[nrow] = nnz)
        return rocprim::lower_bound(temp, temp_bytes, (const uint32_t*)keys, rocprim::counting_iterator<uint32_t>(0),
                                    prow, (size_t)nnz, (size_t)nrow + 1, rocprim::less<uint32_t>(), stream);
    }
    if (L.mode == 2) {
        double* px = (double*)((char*)persist + L.vals_off);
        int32_t* pr = (int32_t*)((char*)persist + L.rows_off);
        int32_t* boff = (int32_t*)((char*)persist + L.boff_off);
        int32_t* table = (int32_t*)((char*)scratch + L.table_off);
        const size_t table_entries = (size_t)L.nsuper * (size_t)L.nblocks + 1;
        e = hipMemsetAsync(table + (table_entries - 1), 0, 4, stream);   // the slot that receives the total
        if (e != hipSuccess) return e;
        const size_t part_lds = (size_t)kTileElems * 12 + ((size_t)L.nblocks * (2 + kPartWaves) + 1) * 4;
        static bool raised = false;   // (benign if two threads both do it)
        if (!raised) {
            e = hipFuncSetAttribute((const void*)rows_tile_partition_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)((size_t)kTileElems * 12 + ((size_t)kPartMaxBlocks * (2 + kPartWaves) + 1) * 4));
            if (e != hipSuccess) return e;
            raised = true;
        }
        if (L.nsuper > 0) {
            hipLaunchKernelGGL(rows_tile_histogram_kernel, dim3(L.nsuper), dim3(kPartThreads),
                               (size_t)L.nblocks * 4, stream, d_i, nnz, nrow, L.shift, L.nblocks, L.super_elems,
                               L.nsuper, table);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        e = rocprim::exclusive_scan(temp, temp_bytes, (const int32_t*)table, table, 0, table_entries,
                                    rocprim::plus<int32_t>(), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(rows_tile_offsets_kernel, dim3((L.nblocks + 1 + 255) / 256), dim3(256), 0, stream, table,
                           L.nblocks, L.nsuper, boff);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (L.nsuper > 0) {
            hipLaunchKernelGGL(rows_tile_partition_kernel, dim3(L.nsuper), dim3(kPartThreads), part_lds, stream, d_x,
                               d_i, nnz, nrow, L.shift, L.nblocks, L.super_elems, L.nsuper, table, px, pr);
            e = hipGetLastError();
        }
        return e;
    }
    double* px = (double*)((char*)persist + L.vals_off);
    uint32_t* pr = (uint32_t*)((char*)persist + L.rows_off);
    int32_t* boff = (int32_t*)((char*)persist + L.boff_off);
    if (nnz > 0 && block_begin_bit(nrow) < key_bits(nrow)) {
        // stable, on the block bits only: the entries of a block keep their storage order
        e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)d_i, pr, d_x, px, (size_t)nnz,
                                      block_begin_bit(nrow), key_bits(nrow), stream);
        if (e != hipSuccess) return e;
    } else if (nnz > 0) {   // a single block: the entries are already together
        e = hipMemcpyAsync(px, d_x, (size_t)nnz * 8, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(pr, d_i, (size_t)nnz * 4, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    temp_bytes = L.temp_bytes;
    // boff[b] = first position whose block id is >= b  (b = 0..nblocks).  Entries whose row index is not
    // in [0, nrow) -- not a valid dgCMatrix -- may sit anywhere; the accumulate kernel leaves them out.
    auto ids = rocprim::make_transform_iterator((const uint32_t*)pr, RowBlockOf{L.shift});
    return rocprim::lower_bound(temp, temp_bytes, ids, rocprim::counting_iterator<uint32_t>(0), boff, (size_t)nnz,
                                (size_t)L.nblocks + 1, rocprim::less<uint32_t>(), stream);
}

// Row sums / means from the row-wise form in `persist`.
hipError_t launch_row_reduce(int32_t nrow, int64_t nnz, const RowSumsLayout& L, void* persist, double* d_out,
                             double divisor, bool means, const LaunchPlan& colsums_plan, hipStream_t stream) {
    if (nrow <= 0) return hipSuccess;
    if (L.mode == 1)   // rowSums(A) = columnSums(t(A)): same kernels, row offsets in place of p
        return launch_column_sums((const double*)((char*)persist + L.vals_off),
                                  (const int32_t*)((char*)persist + L.prow_off), nrow, (int32_t)nnz, d_out,
                                  colsums_plan, (char*)persist + L.colsums_off, divisor, means, stream);
    const double* px = (const double*)((char*)persist + L.vals_off);
    const int32_t* pr = (const int32_t*)((char*)persist + L.rows_off);
    const int32_t* boff = (const int32_t*)((char*)persist + L.boff_off);
    if (L.mode == 2) {
        const size_t acc_lds = ((size_t)8 << L.shift) + (size_t)kAccStage * 12;
        static bool raised2 = false;
        if (!raised2) {
            hipError_t e = hipFuncSetAttribute((const void*)rows_tile_accumulate_kernel<true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)rows_tile_accumulate_kernel<false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
            if (e != hipSuccess) return e;
            raised2 = true;
        }
        if (means)
            hipLaunchKernelGGL(rows_tile_accumulate_kernel<true>, dim3(L.nblocks), dim3(kAccThreads), acc_lds, stream,
                               px, pr, boff, nrow, L.shift, d_out, divisor);
        else
            hipLaunchKernelGGL(rows_tile_accumulate_kernel<false>, dim3(L.nblocks), dim3(kAccThreads), acc_lds, stream,
                               px, pr, boff, nrow, L.shift, d_out, divisor);
        return hipGetLastError();
    }
    const size_t lds = (size_t)8 << L.shift;
    if (means)
        hipLaunchKernelGGL(rows_block_accumulate_kernel<true>, dim3(L.nblocks), dim3(64), lds, stream, px, pr, boff,
                           nrow, L.shift, d_out, divisor);
    else
        hipLaunchKernelGGL(rows_block_accumulate_kernel<false>, dim3(L.nblocks), dim3(64), lds, stream, px, pr, boff,
                           nrow, L.shift, d_out, divisor);
    return hipGetLastError();
}

}  // namespace rsp

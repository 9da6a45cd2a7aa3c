// inspect_device.hip -- the inspectors of the inspector-executor forms of columnSums, ON THE DEVICE.
//
// inspect.hpp (pure host C++) stays the definition: for offsets that already live in HBM these kernels produce
// the same images bit for bit -- the snapped form's records {last column starting at xs0, xs0} per chunk of the
// grid, the lean form's headers {first column, columns} and 16-bit column starts per chunk at the same stride --
// and the statistics the host's choice of form rests on (reference inst/include/RcppSparse.h:220-221: column c
// is [p[c], p[c+1]); nothing here touches x).  Everything is enqueued on the caller's stream: no copy of p[] to
// the host, no synchronisation; the statistics travel to a page-locked host record behind the kernels and the
// host looks at them when they have arrived (capi.hip plan_poll).
//
// Instead of one binary search per chunk (log2(ncol) dependent round trips) the inspection is ONE pass over p[]
// with a thread per column index: column c is the first column at or after a grid position g exactly when
// p[c-1] < g <= p[c], so the thread that sees p[c-1] < p[c] knows every chunk it answers for by two divisions.
//   K1 inspect_columns_kernel   thread c in [0, ncol]: validity of p[], column lengths (min, max), the snapped
//                               records of the planned chunk grid + the largest skip, the lean grid's first columns
//   K2 inspect_lean_chunks_kernel   thread per lean chunk: columns per chunk (widest), reach past the chunk
//   K3 inspect_lean_image_kernel    16 lanes per lean chunk: header + 16-bit offsets at the stride `widest` gives
//   K4 inspect_finish_kernel        one block: the blocks' partial statistics -> the host's page-locked record
#include <hip/hip_runtime.h>
#include <climits>
#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {
namespace {

// Statistics are maxima over all columns (flags are maxima of 0 / 1, the shortest column is kept as INT_MAX - length).
// No atomics and no zeroed memory: every block reduces its own columns (wavefront shuffle, then LDS) and writes ONE
// partial record; the next kernel on the stream reduces the partials.  (One word that 15 000 wavefronts update or
// even just read through L2 costs 200 us: the requests queue up at one L2 channel.)
constexpr int kInspectThreads = 256;
constexpr int kMaxBlocksColumns = kInspectMaxBlocksColumns;   // K1 blocks (grid-stride over the column indices)
constexpr int kMaxBlocksChunks = kInspectMaxBlocksChunks;     // K2 blocks: K3's blocks reduce these partials with one load per thread
static_assert(kMaxBlocksChunks <= kInspectThreads, "one partial per thread in K3");

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}
// maximum over the block's threads of each of N values, valid in thread 0 (every thread has to call)
template <int N>
__device__ __forceinline__ void block_max(int (&v)[N], int* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        v[k] = wave_max(v[k]);
        if (lane == 0) lds[wave * N + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < kInspectThreads / 64; ++w)
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = lds[w * N + k] > v[k] ? lds[w * N + k] : v[k];
}

__global__ __launch_bounds__(kInspectThreads) void inspect_columns_kernel(const int32_t* __restrict__ p, int32_t ncol,
                                                                          int32_t nnz, inspect::Grid grid,
                                                                          int2* __restrict__ rec, inspect::Grid lgrid,
                                                                          int32_t* __restrict__ lean_first,
                                                                          int4* __restrict__ partial) {
    __shared__ int lds[kInspectThreads / 64 * 4];
    int st[4] = {0, 0, 0, 0};   // invalid, max_skip, max_len, INT_MAX - min_len
    const int64_t stride = (int64_t)gridDim.x * kInspectThreads;
    for (int64_t c64 = (int64_t)blockIdx.x * kInspectThreads + threadIdx.x; c64 <= (int64_t)ncol; c64 += stride) {
        const int c = (int)c64;
        const int v = p[c];
        const int prev = c > 0 ? p[c - 1] : -1;
        const int next = c < ncol ? p[c + 1] : INT_MAX;
        // what a dgCMatrix guarantees (and rsp_column_sums_plan_create checks on the host): p[0] = 0, non-decreasing, p[ncol] = nnz
        const bool bad = (c == 0 && v != 0) || v < prev || (c == ncol && v != nnz) || v < 0 || v > nnz;
        if (bad) st[0] = 1;
        if (c < ncol && !bad && next >= v) {
            const int len = next - v;
            st[2] = len > st[2] ? len : st[2];
            st[3] = INT_MAX - len > st[3] ? INT_MAX - len : st[3];
        }
        if (!bad && (c == 0 || prev < v)) {
            // grid positions g with prev < g <= v: this column is the first at or after them, and no column start lies in [g, v)
            const inspect::Span s = inspect::span_of(grid, prev, v);
            if (s.lo <= s.hi) {
                const int64_t sk = (int64_t)v - grid.start(s.lo);
                const int skip = sk > INT_MAX ? INT_MAX : (int)sk;
                st[1] = skip > st[1] ? skip : st[1];
                // the record names the LAST column starting at v (empty columns there end where the previous chunk ends)
                const int last = next == v ? inspect::run_end(p, c, ncol, v) : c;
                // (more than one chunk per column start means a column longer than a chunk: max_skip says "not snapped",
                // and the records of such a plan are never used; a few are written so that short runs stay exact)
                for (int w = s.lo; w <= s.hi && w < s.lo + inspect::kSpanWrites; ++w) rec[w] = make_int2(last, v);
            }
            if (lean_first != nullptr) {
                const inspect::Span l = inspect::span_of(lgrid, prev, v);
                for (int w = l.lo; w <= l.hi && w < l.lo + inspect::kSpanWrites; ++w) lean_first[w] = c;   // (lower_bound: the first of a run)
            }
        }
        if (c == ncol) {
            rec[grid.nchunks] = make_int2(ncol, nnz);
            if (lean_first != nullptr) lean_first[lgrid.nchunks] = ncol;
        }
    }
    block_max(st, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = make_int4(st[0], st[1], st[2], st[3]);
}

// chunk w of the lean grid owns the columns [first[w], first[w + 1]); it has its own rows of x and one more
__global__ __launch_bounds__(kInspectThreads) void inspect_lean_chunks_kernel(const int32_t* __restrict__ p, int32_t ncol,
                                                                              int32_t lean_chunk, int32_t lean_chunks,
                                                                              const int32_t* __restrict__ lean_first,
                                                                              int32_t capacity, int2* __restrict__ partial) {
    __shared__ int lds[kInspectThreads / 64 * 2];
    int st[2] = {0, 0};   // lean_bad, widest
    for (int64_t w = (int64_t)blockIdx.x * kInspectThreads + threadIdx.x; w < lean_chunks; w += (int64_t)gridDim.x * kInspectThreads) {
        int c0 = lean_first[w], c1 = lean_first[w + 1];
        c0 = c0 < 0 ? 0 : (c0 > ncol ? ncol : c0);           // (only an invalid p[] leaves anything to clamp)
        c1 = c1 < c0 ? c0 : (c1 > ncol ? ncol : c1);
        const int n = c1 - c0;
        if (n > 0 && (int64_t)p[c1] - w * lean_chunk > (int64_t)lean_chunk + kRowElems) st[0] = 1;
        if (n > capacity) st[0] = 1;                         // (the image has no room for this chunk's offsets)
        st[1] = n > st[1] ? n : st[1];
    }
    block_max(st, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = make_int2(st[0], st[1]);
}

// the image inspect_lean writes: headers {first column, columns}, then the columns' starts relative to the chunk's
// grid position as 16-bit numbers at a fixed stride (stride from the widest chunk, as on the host)
constexpr int kImageLanes = 16;   // lanes per chunk: each writes one dword (two offsets) per step
__global__ __launch_bounds__(kInspectThreads) void inspect_lean_image_kernel(const int32_t* __restrict__ p, int32_t ncol,
                                                                             int32_t lean_chunk, int32_t lean_chunks,
                                                                             const int32_t* __restrict__ lean_first,
                                                                             int32_t capacity_stride,
                                                                             const int2* __restrict__ chunk_partial,
                                                                             int32_t nparts, int2* __restrict__ hdr,
                                                                             uint32_t* __restrict__ offs) {
    __shared__ int lds[kInspectThreads / 64];
    __shared__ int s_widest;
    int wd[1] = {(int)threadIdx.x < nparts ? chunk_partial[threadIdx.x].y : 0};   // (nparts <= kInspectThreads)
    block_max(wd, lds);
    if (threadIdx.x == 0) s_widest = wd[0];
    __syncthreads();
    int stride = inspect::lean_stride_dwords(s_widest);
    if (stride > capacity_stride) stride = capacity_stride;   // (then lean_bad is set and nobody reads the image)
    const int sub = threadIdx.x & (kImageLanes - 1);
    const int64_t w = ((int64_t)blockIdx.x * kInspectThreads + threadIdx.x) / kImageLanes;
    if (w >= lean_chunks) return;
    int c0 = lean_first[w], c1 = lean_first[w + 1];
    c0 = c0 < 0 ? 0 : (c0 > ncol ? ncol : c0);
    c1 = c1 < c0 ? c0 : (c1 > ncol ? ncol : c1);
    const int n = c1 - c0;
    if (sub == 0) hdr[w] = make_int2(c0, n);
    const int64_t cs = w * lean_chunk;
    uint32_t* mine = offs + (size_t)w * (size_t)stride;
    for (int d = sub; d < stride; d += kImageLanes) {
        uint32_t lo = 0, hi = 0;
        if (n > 0) {                                          // (a chunk without columns keeps an all-zero row)
            if (2 * d <= n) lo = (uint32_t)(uint16_t)((int64_t)p[c0 + 2 * d] - cs);
            if (2 * d + 1 <= n) hi = (uint32_t)(uint16_t)((int64_t)p[c0 + 2 * d + 1] - cs);
        }
        mine[d] = lo | (hi << 16);
    }
}

// one block: the partials of K1 (and K2) -> the statistics, written straight into the host's page-locked record
__global__ __launch_bounds__(kInspectThreads) void inspect_finish_kernel(const int4* __restrict__ col_partial, int32_t ncolparts,
                                                                         const int2* __restrict__ chunk_partial,
                                                                         int32_t nchunkparts, PlanStats* __restrict__ out) {
    __shared__ int lds[kInspectThreads / 64 * 6];
    int st[6] = {0, 0, 0, 0, 0, 0};
    for (int k = threadIdx.x; k < ncolparts; k += kInspectThreads) {
        const int4 q = col_partial[k];
        st[0] = q.x > st[0] ? q.x : st[0];
        st[1] = q.y > st[1] ? q.y : st[1];
        st[2] = q.z > st[2] ? q.z : st[2];
        st[3] = q.w > st[3] ? q.w : st[3];
    }
    for (int k = threadIdx.x; k < nchunkparts; k += kInspectThreads) {
        const int2 q = chunk_partial[k];
        st[4] = q.x > st[4] ? q.x : st[4];
        st[5] = q.y > st[5] ? q.y : st[5];
    }
    block_max(st, lds);
    if (threadIdx.x == 0) {
        PlanStats r{};
        r.invalid = st[0];
        r.max_skip = st[1];
        r.max_len = st[2];
        r.inv_min_len = st[3];
        r.lean_bad = st[4];
        r.lean_widest = st[5];
        *out = r;
        // the host may look at the page-locked record at any moment (capi.hip plan_poll: no event query, no wait): the
        // ready word goes out only after the statistics have
        __threadfence_system();
        __hip_atomic_store(&out->ready, inspect::kStatsReady, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

// Enqueues the inspection of d_p on `stream`; the statistics are written to `stats_out`, which may be (and in capi.hip is)
// page-locked host memory: the host reads it once an event recorded behind this call has completed.
hipError_t launch_inspect_device(const int32_t* d_p, int32_t ncol, int32_t nnz, const LaunchPlan& grid,
                                 const DeviceInspectLayout& L, void* d_mem, PlanStats* stats_out, hipStream_t stream) {
    char* base = (char*)d_mem;
    int2* rec = (int2*)(base + L.rec_off);
    int32_t* first = L.try_lean ? (int32_t*)(base + L.first_off) : nullptr;
    int4* part1 = (int4*)(base + L.part1_off);
    int2* part2 = (int2*)(base + L.part2_off);
    const inspect::Grid g{grid.chunk_elems, grid.nbody, grid.tail_elems, grid.nchunks};
    const int32_t lchunk = L.lean_rows * kRowElems;
    const inspect::Grid lg{lchunk, L.lean_chunks, lchunk, L.lean_chunks};
    const int64_t want1 = ((int64_t)ncol + 1 + kInspectThreads - 1) / kInspectThreads;
    const int nb1 = (int)(want1 < kMaxBlocksColumns ? want1 : kMaxBlocksColumns);
    hipLaunchKernelGGL(inspect_columns_kernel, dim3(nb1), dim3(kInspectThreads), 0, stream, d_p, ncol, nnz, g, rec, lg,
                       first, part1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int nb2 = 0;
    if (L.try_lean) {
        int2* hdr = (int2*)(base + L.hdr_off);
        uint32_t* offs = (uint32_t*)(hdr + L.lean_chunks);
        const int64_t want2 = ((int64_t)L.lean_chunks + kInspectThreads - 1) / kInspectThreads;
        nb2 = (int)(want2 < kMaxBlocksChunks ? want2 : kMaxBlocksChunks);
        hipLaunchKernelGGL(inspect_lean_chunks_kernel, dim3(nb2), dim3(kInspectThreads), 0, stream, d_p, ncol, lchunk,
                           L.lean_chunks, first, L.lean_capacity, part2);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int64_t lanes = (int64_t)L.lean_chunks * kImageLanes;
        hipLaunchKernelGGL(inspect_lean_image_kernel, dim3((unsigned)((lanes + kInspectThreads - 1) / kInspectThreads)),
                           dim3(kInspectThreads), 0, stream, d_p, ncol, lchunk, L.lean_chunks, first,
                           L.lean_capacity_stride, part2, nb2, hdr, offs);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(inspect_finish_kernel, dim3(1), dim3(kInspectThreads), 0, stream, part1, nb1, part2, nb2, stats_out);
    return hipGetLastError();
}

}  // namespace rsp

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
#include <hip/hip_runtime.h>
#include <climits>
#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {
namespace {

// Statistics are maxima / minima over all columns: a wavefront combines its lanes first and only then, and only
// if its value would change the word, sends one atomic -- a few atomics per word and launch instead of one per
// wavefront (15 000 of them on one address cost more than the pass itself).
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ void stat_max(int32_t* word, int v, int lane) {
    v = wave_max(v);
    if (lane == 0 && v > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, v);
}

__global__ __launch_bounds__(256) void inspect_columns_kernel(const int32_t* __restrict__ p, int32_t ncol, int32_t nnz,
                                                              inspect::Grid grid, int2* __restrict__ rec,
                                                              inspect::Grid lgrid,
                                                              int32_t* __restrict__ lean_first,
                                                              PlanStats* __restrict__ st) {
    const int lane = threadIdx.x & 63;
    const int64_t c64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c64 <= (int64_t)ncol;
    const int c = live ? (int)c64 : ncol;
    const int v = p[c];
    const int prev = c > 0 ? p[c - 1] : -1;
    const int next = c < ncol ? p[c + 1] : INT_MAX;
    // what a dgCMatrix guarantees (and rsp_column_sums_plan_create checks on the host): p[0] = 0, non-decreasing, p[ncol] = nnz
    const bool bad = live && ((c == 0 && v != 0) || v < prev || (c == ncol && v != nnz) || v < 0 || v > nnz);
    if (__ballot(bad) != 0ull && lane == 0) st->invalid = 1;
    const bool column = live && c < ncol && !bad && next >= v;
    const int len = column ? next - v : 0;
    stat_max(&st->max_len, len, lane);
    stat_max(&st->inv_min_len, column ? INT_MAX - len : 0, lane);   // (a minimum kept as a maximum: one memset to zero starts every word)

    const bool run_first = live && !bad && (c == 0 || prev < v);
    int skip = 0;
    if (run_first) {
        // grid positions g with prev < g <= v: this column is the first at or after them, and no column start lies in [g, v)
        const inspect::Span s = inspect::span_of(grid, prev, v);
        if (s.lo <= s.hi) {
            const int64_t sk = (int64_t)v - grid.start(s.lo);
            skip = sk > INT_MAX ? INT_MAX : (int)sk;
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
    stat_max(&st->max_skip, skip, lane);
    if (live && c == ncol) {
        rec[grid.nchunks] = make_int2(ncol, nnz);
        if (lean_first != nullptr) lean_first[lgrid.nchunks] = ncol;
    }
}

// chunk w of the lean grid owns the columns [first[w], first[w + 1]); it has its own rows of x and one more
__global__ __launch_bounds__(256) void inspect_lean_chunks_kernel(const int32_t* __restrict__ p, int32_t ncol,
                                                                  int32_t lean_chunk, int32_t lean_chunks,
                                                                  const int32_t* __restrict__ lean_first,
                                                                  int32_t capacity, PlanStats* __restrict__ st) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    int n = 0;
    bool bad = false;
    if (w < lean_chunks) {
        int c0 = lean_first[w], c1 = lean_first[w + 1];
        c0 = c0 < 0 ? 0 : (c0 > ncol ? ncol : c0);           // (only an invalid p[] leaves anything to clamp)
        c1 = c1 < c0 ? c0 : (c1 > ncol ? ncol : c1);
        n = c1 - c0;
        if (n > 0 && (int64_t)p[c1] - (int64_t)w * lean_chunk > (int64_t)lean_chunk + kRowElems) bad = true;
        if (n > capacity) bad = true;                        // (the image has no room for this chunk's offsets)
    }
    if (__ballot(bad) != 0ull && lane == 0) st->lean_bad = 1;
    stat_max(&st->lean_widest, n, lane);
}

// the image inspect_lean writes: headers {first column, columns}, then the columns' starts relative to the chunk's
// grid position as 16-bit numbers at a fixed stride (stride from the widest chunk, as on the host)
constexpr int kImageLanes = 16;   // lanes per chunk: each writes one dword (two offsets) per step
__global__ __launch_bounds__(256) void inspect_lean_image_kernel(const int32_t* __restrict__ p, int32_t ncol,
                                                                 int32_t lean_chunk, int32_t lean_chunks,
                                                                 const int32_t* __restrict__ lean_first,
                                                                 int32_t capacity_stride,
                                                                 const PlanStats* __restrict__ st,
                                                                 int2* __restrict__ hdr, uint32_t* __restrict__ offs) {
    const int widest = st->lean_widest;
    int stride = inspect::lean_stride_dwords(widest);
    if (stride > capacity_stride) stride = capacity_stride;   // (then lean_bad is set and nobody reads the image)
    const int sub = threadIdx.x & (kImageLanes - 1);
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) / kImageLanes;
    if (w >= lean_chunks) return;
    int c0 = lean_first[w], c1 = lean_first[w + 1];
    c0 = c0 < 0 ? 0 : (c0 > ncol ? ncol : c0);
    c1 = c1 < c0 ? c0 : (c1 > ncol ? ncol : c1);
    const int n = c1 - c0;
    if (sub == 0) hdr[w] = make_int2(c0, n);
    const int64_t cs = (int64_t)w * lean_chunk;
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

}  // namespace

hipError_t launch_inspect_device(const int32_t* d_p, int32_t ncol, int32_t nnz, const LaunchPlan& grid,
                                 const DeviceInspectLayout& L, void* d_mem, hipStream_t stream) {
    char* base = (char*)d_mem;
    PlanStats* st = (PlanStats*)(base + L.stats_off);
    int2* rec = (int2*)(base + L.rec_off);
    int32_t* first = L.try_lean ? (int32_t*)(base + L.first_off) : nullptr;
    hipError_t e = hipMemsetAsync(st, 0, sizeof(PlanStats), stream);   // every statistic is a flag or a maximum from zero
    if (e != hipSuccess) return e;
    const inspect::Grid g{grid.chunk_elems, grid.nbody, grid.tail_elems, grid.nchunks};
    const int32_t lchunk = L.lean_rows * kRowElems;
    const inspect::Grid lg{lchunk, L.lean_chunks, lchunk, L.lean_chunks};
    const int64_t threads = (int64_t)ncol + 1;
    hipLaunchKernelGGL(inspect_columns_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_p, ncol,
                       nnz, g, rec, lg, first, st);
    e = hipGetLastError();
    if (e != hipSuccess || !L.try_lean) return e;
    int2* hdr = (int2*)(base + L.hdr_off);
    uint32_t* offs = (uint32_t*)(hdr + L.lean_chunks);
    hipLaunchKernelGGL(inspect_lean_chunks_kernel, dim3((L.lean_chunks + 255) / 256), dim3(256), 0, stream, d_p, ncol,
                       L.lean_rows * kRowElems, L.lean_chunks, first, L.lean_capacity, st);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t lanes = (int64_t)L.lean_chunks * kImageLanes;
    hipLaunchKernelGGL(inspect_lean_image_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, d_p, ncol,
                       L.lean_rows * kRowElems, L.lean_chunks, first, L.lean_capacity_stride, st, hdr, offs);
    return hipGetLastError();
}

}  // namespace rsp

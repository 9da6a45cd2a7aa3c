// scan.hip -- exclusive prefix sum of 32-bit counts, hand-written (round 4: the last library call of the row-wise
// paths; rounds 1-3 called rocPRIM here).  Used on the count tables of the regrouping passes: entries per (row block,
// supertile) of Matrix::rowSums (rowsums.hip) and entries per row of crossprod's row-major form (crossprod.hip) -- a few
// MB, off the hot path.  Three launches over tiles of 4096 counts: every block adds up its tile; ONE block scans the
// tiles' totals (carrying from chunk to chunk of 1024); every block scans its tile again on top of its total.  In place
// or out of place; 12 B of traffic per count.  Integer adds: the result does not depend on any order.
// Round 5: up to kScanSelfTiles tiles the middle launch is dropped -- every block of the last pass adds up the totals of
// the tiles before it for itself (a few KB out of L2); an optional second copy of the result (crossprod's cursors: no
// device-to-device copy behind the scan); an optional device flag on which all launches stand by (run_if: crossprod's
// exact kernels behind the matrix-core form do nothing at all unless a sum was not finite).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {
namespace {

constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanTile = kScanThreads * kScanPerThread;   // 4096 counts per block

__device__ __forceinline__ int32_t wave_inclusive_scan(int32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread over the block; *total = the block's sum (in every thread)
__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t* lds, int32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        const int32_t t = lds[w];
        if (w < wave) base += t;
        all += t;
    }
    __syncthreads();   // (lds is reused by the caller's next round)
    *total = all;
    return base + inc - v;
}

__device__ __forceinline__ void load_tile(const int32_t* __restrict__ in, int64_t n, int64_t first, int32_t (&v)[kScanPerThread]) {
    const int64_t e0 = first + (int64_t)threadIdx.x * kScanPerThread;
    if (e0 + kScanPerThread <= n && (((uintptr_t)(in + e0)) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < kScanPerThread / 4; ++k) {
            const int4 q = *reinterpret_cast<const int4*>(in + e0 + 4 * k);
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) v[k] = e0 + k < n ? in[e0 + k] : 0;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_tile_totals_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                        int32_t* __restrict__ totals,
                                                                        const int32_t* __restrict__ run_if) {
    __shared__ int32_t lds[kScanThreads / 64];
    if (run_if && *run_if == 0) return;
    int32_t v[kScanPerThread];
    load_tile(in, n, (int64_t)blockIdx.x * kScanTile, v);
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) s += v[k];
    int32_t total;
    (void)block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// one block: totals[b] := sum of totals[0 .. b)
__global__ __launch_bounds__(1024) void scan_totals_kernel(int32_t* __restrict__ totals, int32_t nb,
                                                           const int32_t* __restrict__ run_if) {
    __shared__ int32_t lds[16];
    __shared__ int32_t s_carry;
    if (run_if && *run_if == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int32_t b0 = 0; b0 < nb; b0 += 1024) {
        const int32_t b = b0 + threadIdx.x;
        const int32_t v = b < nb ? totals[b] : 0;
        const int32_t inc = wave_inclusive_scan(v, lane);
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        int32_t base = s_carry, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int32_t t = lds[w];
            if (w < wave) base += t;
            all += t;
        }
        if (b < nb) totals[b] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += all;
        __syncthreads();
    }
}

// SELF: `totals` holds the tiles' sums as scan_tile_totals_kernel left them, and this block adds up those before its own
template <bool SELF>
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(const int32_t* in, int64_t n,   // (in may be out or out2)
                                                                  const int32_t* __restrict__ totals, int32_t initial,
                                                                  int32_t* out, int32_t* out2,
                                                                  const int32_t* __restrict__ run_if) {
    __shared__ int32_t lds[kScanThreads / 64];
    if (run_if && *run_if == 0) return;
    int32_t v[kScanPerThread];
    const int64_t first = (int64_t)blockIdx.x * kScanTile;
    load_tile(in, n, first, v);       // (in place: a block reads its whole tile before it writes any of it)
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) s += v[k];
    int32_t before = 0, total;
    if (SELF) {
        int32_t mine = 0;
        for (int32_t b = threadIdx.x; b < (int32_t)blockIdx.x; b += kScanThreads) mine += totals[b];
        (void)block_exclusive_scan(mine, lds, &before);
    } else {
        before = totals[blockIdx.x];
    }
    int32_t run = initial + before + block_exclusive_scan(s, lds, &total);
    const int64_t e0 = first + (int64_t)threadIdx.x * kScanPerThread;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const int32_t x = v[k];
        if (e0 + k < n) {
            out[e0 + k] = run;
            if (out2) out2[e0 + k] = run;
        }
        run += x;
    }
}

constexpr int64_t kScanSelfTiles = 2048;   // (8.4 million counts: the last block reads 8 KB of totals, eight loads per thread;
                                           // at 8192 tiles the blocks' own sums would cost what the one-block pass does)

}  // namespace

size_t exclusive_scan_temp_bytes(int64_t n) {
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    return (size_t)(nb > 0 ? nb : 1) * 4 + 256;
}

// out[k] = initial + in[0] + ... + in[k - 1], k in [0, n); `out` may be `in`.  temp: exclusive_scan_temp_bytes(n) bytes.
// out2 (optional; may be `in`, not `out`): a second copy of the result.  run_if (optional): a device word; if it is 0
// when the launches run, they do nothing.
hipError_t launch_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t initial, void* temp,
                                     size_t temp_bytes, hipStream_t stream, const int32_t* run_if, int32_t* out2) {
    if (n <= 0) return hipSuccess;
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    if (nb > 0x7fffffff || temp_bytes < (size_t)nb * 4) return hipErrorInvalidValue;
    int32_t* totals = (int32_t*)temp;
    hipLaunchKernelGGL(scan_tile_totals_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, n, totals, run_if);
    if (nb <= kScanSelfTiles) {
        hipLaunchKernelGGL(scan_tiles_kernel<true>, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, n,
                           (const int32_t*)totals, initial, out, out2, run_if);
    } else {
        hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, stream, totals, (int32_t)nb, run_if);
        hipLaunchKernelGGL(scan_tiles_kernel<false>, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, n,
                           (const int32_t*)totals, initial, out, out2, run_if);
    }
    return hipGetLastError();
}

}  // namespace rsp

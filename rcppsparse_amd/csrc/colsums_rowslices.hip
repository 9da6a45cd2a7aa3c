// Row-restricted column sums, slice-major form ("next" row f4; reference RcppSparse.h:238-321,
// Matrix::InnerIteratorInRange / InnerIteratorNotInRange: a column's entries whose row is / is not in a set).
//
// The general masked kernel (colsums_kernels.hip) probes the row bitmap once per entry.  Up to 128 KB the bitmap
// sits in LDS; above that (more than 2^20 rows) every probe is a 4-byte gather that moves a 128-byte line from L2,
// and the call is bound by L2's line rate at 25 % of its 12 B/nnz roofline (profiles/r03_masked.md).
//
// This form uses what a dgCMatrix guarantees: inside a column the rows ascend, so the entries of column c that
// fall into the row slice [s * 2^20, (s + 1) * 2^20) are CONTIGUOUS in x / i.  A workgroup owns a group of
// columns and walks the slices in order; per slice it copies that slice's 128 KB of bitmap into LDS once, and
// every column keeps a cursor (LDS) that stands where the previous slice ended -- there is no search: the piece of
// i[] a wavefront loads at the cursor tells it how many of the entries belong to the slice (a prefix), exactly
// those entries of x are requested, their bits are probed in LDS, and the piece's sum is added to the column's
// accumulator (LDS).  x is read once, i once plus what a piece reads past its slice's end; no workspace, no
// carries, one launch.  Column c's result = its slices' piece sums added in slice order (= storage order), each
// piece a fixed 64-lane tree: deterministic, within the tolerance of the general form, not its bits.
//
// It pays when a column has tens of entries per slice (C3 shape: 1000 entries over 10 slices); the launcher
// selects it by nnz / (ncol * slices) and a device-side guard hands matrices with giant columns or unbalanced
// column groups back to the general kernel (one wavefront walks a whole segment here).
#include "colsums_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

namespace rsp {
namespace {

#ifndef RSP_SLC_NT
#define RSP_SLC_NT 3   // bit 0: row indices, bit 1: values loaded with the nontemporal hint
#endif
template <class T>
__device__ __forceinline__ T slc_load_i(const T* q) { return (RSP_SLC_NT & 1) ? __builtin_nontemporal_load(q) : *q; }
template <class T>
__device__ __forceinline__ T slc_load_x(const T* q) { return (RSP_SLC_NT & 2) ? __builtin_nontemporal_load(q) : *q; }

constexpr int kSlcWaves = 16;                 // one workgroup per CU: its LDS is the slice's bitmap
constexpr int kSlcThreads = kSlcWaves * 64;
constexpr int kSlcBatch = 8;                  // columns a wavefront has in flight
constexpr int kSlcWords = 1 << (kSliceRowsShift - 5);   // 32768 words = 128 KB
constexpr int kSlcPiece = 128;                // entries of one column a wavefront takes per round (2 per lane)

template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor_lanes(double v, int mask, int lane) {   // lanes further apart than a DPP row
    const int src = (lane ^ mask) << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(v)),
                            __builtin_amdgcn_ds_bpermute(src, __double2loint(v)));
}

// t[j] = lane's term of column j.  Returns, in lane l, the sum over all 64 lanes of column l & 7: three
// halving steps (8 -> 4 -> 2 -> 1 values per lane: a lane keeps the half its lane bit names and receives its
// partner's copy of the same half), then three plain steps over lanes that hold the same column.  Fixed tree.
__device__ __forceinline__ double reduce8(const double (&t)[kSlcBatch], int lane) {
    const bool b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    double w[4], u[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {   // partner 7 - l (row_half_mirror) has the other value of bit 2
        const double keep = b2 ? t[m + 4] : t[m], send = b2 ? t[m] : t[m + 4];
        w[m] = keep + dppd<0x141>(send);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {   // partner l ^ 2
        const double keep = b1 ? w[m + 2] : w[m], send = b1 ? w[m] : w[m + 2];
        u[m] = keep + dppd<0x4E>(send);
    }
    const double keep = b0 ? u[1] : u[0], send = b0 ? u[0] : u[1];
    double r = keep + dppd<0xB1>(send);          // partner l ^ 1: column (l & 7) over the lane's group of 8
    r += dppd<0x128>(r);                         // row_ror:8 = l ^ 8 inside a row of 16
    r += xor_lanes(r, 16, lane);
    r += xor_lanes(r, 32, lane);
    return r;
}

struct Batch {   // 8 columns of one wavefront: cursors and column ends (wave-uniform), the piece's row indices (2 per lane)
    int cur[kSlcBatch], pend[kSlcBatch];
    int r0[kSlcBatch], r1[kSlcBatch];
    uint32_t have0, have1;   // bit j: this lane has loaded r0[j] / r1[j] (no sentinel row value: every 32-bit pattern is data)
};

template <bool COMPLEMENT>
__global__ __launch_bounds__(kSlcThreads) void colsums_rowslices_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, const int32_t* __restrict__ p, int32_t ncol,
    int32_t nnz, int32_t nslices, const uint32_t* __restrict__ bitmap, int32_t bitmap_words, int32_t group,
    double* __restrict__ out, const int32_t* __restrict__ skip_if) {
    if (skip_if != nullptr && *skip_if != 0) return;   // (the guard's verdict: the general kernel runs instead)
    extern __shared__ uint32_t s_lds[];
    uint32_t* s_bits = s_lds;                                            // kSlcWords
    int32_t* s_cur = reinterpret_cast<int32_t*>(s_lds + kSlcWords);      // kSliceMaxGroup
    double* s_sum = reinterpret_cast<double*>(s_cur + kSliceMaxGroup);   // kSliceMaxGroup

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c_begin = blockIdx.x * group;
    const int c_n = min(group, ncol - c_begin);
    for (int k = threadIdx.x; k < c_n; k += kSlcThreads) {
        s_cur[k] = min(max(p[c_begin + k], 0), nnz);   // (offsets outside [0, nnz] -- not a valid matrix -- read nothing out of bounds)
        s_sum[k] = 0.0;
    }

    for (int s = 0; s < nslices; ++s) {
        __syncthreads();   // the previous slice's probes are done (and, first time round, the cursors are there)
        const int wbase = s * kSlcWords;
#pragma unroll 8
        for (int k = threadIdx.x; k < kSlcWords; k += kSlcThreads) {
            const int g = wbase + k;
            s_bits[k] = g < bitmap_words ? bitmap[g] : 0u;
        }
        __syncthreads();
        const uint32_t row_base = (uint32_t)s << kSliceRowsShift;
        const uint32_t row_end = row_base + (1u << kSliceRowsShift);   // (<= 2^31; the last slice may end exactly there)
        // The last slice takes whatever is left of a column, also rows at or beyond nrow (not a valid dgCMatrix): their
        // bits lie outside the bitmap and read as "not in the set", which is what the general kernel makes of them.
        const bool last_slice = s == nslices - 1;

        // A wavefront takes its batches of 8 columns in order.  Software pipeline: the row indices of the NEXT
        // batch's first pieces (and the column ends of the one after) are requested before the current batch's
        // values are, so the memory system always has a request of this wavefront queued while it waits.
        constexpr int kStride = kSlcWaves * kSlcBatch;
        auto ends_of = [&](int k0) -> int {   // lane j: end of column k0 + j
            return (lane < kSlcBatch && k0 + lane < c_n) ? min(max(p[c_begin + k0 + lane + 1], 0), nnz) : 0;
        };
        auto first_pieces = [&](int k0, int pe, Batch& b) {   // (past the group's last column: cur = pend = 0, no loads)
#pragma unroll
            for (int j = 0; j < kSlcBatch; ++j) {
                b.cur[j] = __builtin_amdgcn_readfirstlane(k0 + j < c_n ? s_cur[k0 + j] : 0);
                b.pend[j] = __builtin_amdgcn_readlane(pe, j);
            }
            b.have0 = b.have1 = 0u;
#pragma unroll
            for (int j = 0; j < kSlcBatch; ++j) {
                const uint32_t e0 = (uint32_t)b.cur[j] + (uint32_t)lane, e1 = e0 + 64u;   // (unsigned: a cursor may stand at 2^31 - 1)
                b.r0[j] = b.r1[j] = 0;
                if (e0 < (uint32_t)b.pend[j]) { b.r0[j] = slc_load_i(ri + e0); b.have0 |= 1u << j; }
                if (e1 < (uint32_t)b.pend[j]) { b.r1[j] = slc_load_i(ri + e1); b.have1 |= 1u << j; }
            }
        };
        int k0 = wave * kSlcBatch;
        Batch cb;
        first_pieces(k0, ends_of(k0), cb);
        int pe_next = ends_of(k0 + kStride);
        for (; k0 < c_n; k0 += kStride) {
            Batch nb;
            first_pieces(k0 + kStride, pe_next, nb);
            pe_next = ends_of(k0 + 2 * kStride);
            uint32_t more = c_n - k0 >= kSlcBatch ? (1u << kSlcBatch) - 1 : (1u << (c_n - k0)) - 1;
            for (;;) {
                double t[kSlcBatch], x0[kSlcBatch], x1[kSlcBatch];
                int n[kSlcBatch];
#pragma unroll
                for (int j = 0; j < kSlcBatch; ++j) {   // the piece's entries inside the slice are a prefix (rows ascend)
                    const uint32_t e0 = (uint32_t)cb.cur[j] + (uint32_t)lane, e1 = e0 + 64u;
                    const bool v0 = ((more & cb.have0) >> j & 1) && (last_slice || (uint32_t)cb.r0[j] < row_end);
                    const bool v1 = ((more & cb.have1) >> j & 1) && (last_slice || (uint32_t)cb.r1[j] < row_end);
                    n[j] = __builtin_popcountll(__ballot(v0)) + __builtin_popcountll(__ballot(v1));
                    x0[j] = x1[j] = 0.0;
                    if (v0) x0[j] = slc_load_x(x + e0);
                    if (v1) x1[j] = slc_load_x(x + e1);
                }
#pragma unroll
                for (int j = 0; j < kSlcBatch; ++j) {   // (an entry that was not taken has x = 0)
                    const uint32_t q0 = (uint32_t)cb.r0[j] - row_base, q1 = (uint32_t)cb.r1[j] - row_base;
                    const uint32_t w0 = q0 >> 5, w1 = q1 >> 5;
                    const uint32_t m0 = w0 < (uint32_t)kSlcWords ? s_bits[w0] : 0u;   // (a row below the slice -- rows
                    const uint32_t m1 = w1 < (uint32_t)kSlcWords ? s_bits[w1] : 0u;   // not ascending -- probes nothing)
                    const bool in0 = ((m0 >> (q0 & 31)) & 1u) != (COMPLEMENT ? 1u : 0u);
                    const bool in1 = ((m1 >> (q1 & 31)) & 1u) != (COMPLEMENT ? 1u : 0u);
                    t[j] = (in0 ? x0[j] : 0.0) + (in1 ? x1[j] : 0.0);
                }
                const double total = reduce8(t, lane);
                if (lane < kSlcBatch && ((more >> lane) & 1)) s_sum[k0 + lane] += total;
                uint32_t again = 0;
#pragma unroll
                for (int j = 0; j < kSlcBatch; ++j) {
                    cb.cur[j] += n[j];
                    if (((more >> j) & 1) && n[j] == kSlcPiece) again |= 1u << j;   // the whole piece was the slice's: there may be more
                }
                more = again;
                if (more == 0) break;
                cb.have0 = cb.have1 = 0u;
#pragma unroll
                for (int j = 0; j < kSlcBatch; ++j) {   // (long segments: the next piece of the columns that go on)
                    cb.r0[j] = cb.r1[j] = 0;
                    if ((more >> j) & 1) {
                        const uint32_t e0 = (uint32_t)cb.cur[j] + (uint32_t)lane, e1 = e0 + 64u;
                        if (e0 < (uint32_t)cb.pend[j]) { cb.r0[j] = slc_load_i(ri + e0); cb.have0 |= 1u << j; }
                        if (e1 < (uint32_t)cb.pend[j]) { cb.r1[j] = slc_load_i(ri + e1); cb.have1 |= 1u << j; }
                    }
                }
            }
            int mine = 0;
#pragma unroll
            for (int j = 0; j < kSlcBatch; ++j) mine = lane == j ? cb.cur[j] : mine;
            if (lane < kSlcBatch && k0 + lane < c_n) s_cur[k0 + lane] = mine;
            cb = nb;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < c_n; k += kSlcThreads) out[c_begin + k] = s_sum[k];
}

// Device-side guard: sets *flag when a column is longer than max_col entries or a group of columns holds more
// than max_group entries (a single wavefront walks a column's segment; a single workgroup a group).
__global__ __launch_bounds__(256) void colsums_rowslices_guard_kernel(const int32_t* __restrict__ p, int32_t ncol,
                                                                      int32_t group, int32_t max_col,
                                                                      int64_t max_group, int32_t* __restrict__ flag) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int32_t a = p[c];
    bool bad = p[c + 1] - a > max_col;
    if (c % group == 0) bad |= (int64_t)p[min(c + group, ncol)] - a > max_group;
    if (bad) *flag = 1;
}

}  // namespace

bool rowslices_applicable(int32_t nrow, int32_t ncol, int64_t nnz, bool force, RowSlicesPlan* out) {
    if ((int64_t)nrow <= (int64_t)1 << kSliceRowsShift) return false;   // the bitmap fits in LDS: the LDS form of the general kernel
    const int32_t nslices = (int32_t)(((int64_t)nrow + ((int64_t)1 << kSliceRowsShift) - 1) >> kSliceRowsShift);
    if (ncol < 1 || nnz < 1) return false;
    // (force: tests take the form on matrices of any shape; what follows is about speed, not about correctness)
    if (!force && ncol < kSliceMinColumns) return false;
    if (!force && nnz < (int64_t)kSliceMinSegment * ncol * nslices) return false;
    // groups: a whole number of rounds of one workgroup per CU, at most kSliceMaxGroup columns each
    const int64_t per_round = (int64_t)kSliceCus * kSliceMaxGroup;
    const int64_t rounds = (ncol + per_round - 1) / per_round;
    const int64_t units = rounds * kSliceCus;
    int32_t group = (int32_t)((ncol + units - 1) / units);
    group = (group + kSlcBatch - 1) / kSlcBatch * kSlcBatch;
    if (group > kSliceMaxGroup) group = kSliceMaxGroup;
    // a workgroup copies 128 KB of bitmap per slice: its columns' entries in that slice should outweigh the copy
    if (!force && nnz / ((int64_t)ncol * nslices) * group < kSliceMinEntriesPerPass) return false;
    if (out) {
        out->nslices = nslices;
        out->group = group;
        out->ngroups = (int32_t)(((int64_t)ncol + group - 1) / group);
        const int64_t mean_col = nnz / ncol;
        out->max_col = (int32_t)std::min<int64_t>(INT32_MAX, mean_col * kSliceMaxColumnFactor + 4096);
        out->max_group = 2 * (nnz / out->ngroups) + 65536;
    }
    return true;
}

hipError_t launch_rowslices_guard(const int32_t* d_p, int32_t ncol, const RowSlicesPlan& sp, int32_t* d_flag,
                                  hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(colsums_rowslices_guard_kernel, dim3((ncol + 255) / 256), dim3(256), 0, stream, d_p, ncol,
                       sp.group, sp.max_col, sp.max_group, d_flag);
    return hipGetLastError();
}

hipError_t launch_column_sums_rowslices(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t ncol,
                                        int32_t nnz, const uint32_t* d_bitmap, int32_t bitmap_words, bool complement,
                                        const RowSlicesPlan& sp, double* d_out, const int32_t* d_skip_if,
                                        hipStream_t stream) {
    constexpr int lds = kSlcWords * 4 + kSliceMaxGroup * (4 + 8);
    static DynamicLdsLimit lim_in, lim_out;
    hipError_t e = complement ? lim_out.ensure((const void*)colsums_rowslices_kernel<true>, lds)
                              : lim_in.ensure((const void*)colsums_rowslices_kernel<false>, lds);
    if (e != hipSuccess) return e;
    const dim3 grid(sp.ngroups), block(kSlcThreads);
    if (complement)
        hipLaunchKernelGGL(colsums_rowslices_kernel<true>, grid, block, lds, stream, d_x, d_i, d_p, ncol, nnz,
                           sp.nslices, d_bitmap, bitmap_words, sp.group, d_out, d_skip_if);
    else
        hipLaunchKernelGGL(colsums_rowslices_kernel<false>, grid, block, lds, stream, d_x, d_i, d_p, ncol, nnz,
                           sp.nslices, d_bitmap, bitmap_words, sp.group, d_out, d_skip_if);
    return hipGetLastError();
}

}  // namespace rsp

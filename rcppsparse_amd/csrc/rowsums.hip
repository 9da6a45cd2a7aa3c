// rowsums.hip -- Matrix::rowSums / rowMeans on the device ("next" row f1 of SURVEY.md 8f).
//
// Reference inst/include/RcppSparse.h:138-144 scatters sums(i[j]) += x[j] while walking the
// columns.  On the device that is a reduction by key with 1e7 keys and no locality in the key.
// Two forms, chosen by the caller (capi.hip):
//
//   block form (one-shot calls, rsp_row_sums_device): the (i, x) pairs are sorted by ROW BLOCK only
//     -- 4096 consecutive rows, few enough for one wavefront to keep the block's sums in LDS -- with
//     a stable device radix sort over just the block bits of the row index (rocPRIM; 12 of the 24
//     bits for 1e7 rows: two passes instead of three), and rows_block_accumulate_kernel (hand-written)
//     adds each block's entries into LDS with one ds_add_f64 per 64 entries and writes the block's
//     sums.  The sort is stable, so a row's terms arrive in the reference's ascending storage
//     order; lanes of one LDS instruction that hit the same row are serialised by the hardware.
//
//   row form (the handle API, which keeps it for repeated calls): full stable sort by row,
//     row offsets by a vectorised lower_bound, then the column-sum kernels on the row-major
//     values (8 B/nnz per repeated call instead of 12).
//
// Both are deterministic (no float atomics in global memory, no timing-dependent order) and within
// the usual 1e-12 * sum|x| of the reference.
//
// Measured and NOT kept (profiles/r02_rowsums.md): a single partition pass into ~1200 row blocks
// (per-supertile count table + LDS cursors, hand-written).  Its traffic is only ~40 B/nnz, but with
// ~1200 open output streams per wavefront every store instruction becomes 64 separate 8-byte memory
// transactions: 47 ms for the scatter alone on C3, against 9 ms per radix pass.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kRowBlockShift = 12;   // 4096 rows per block: 32 KB of LDS sums per wavefront, 5 per CU

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

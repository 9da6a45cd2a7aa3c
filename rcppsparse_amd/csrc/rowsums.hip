// rowsums.hip -- Matrix::rowSums / rowMeans on the device ("next" row f1 of SURVEY.md 8f).
//
// Reference inst/include/RcppSparse.h:138-144 scatters sums(i[j]) += x[j] while walking
// the columns, i.e. every row is accumulated in ascending storage order j.  Here
//     rowSums(A) = columnSums(t(A))
// without materialising t(A): a *stable* radix sort of the (i[j], x[j]) pairs by row
// (rocPRIM device radix sort; only the bits that nrow needs) puts each row's values
// next to each other in that same ascending-j order, a vectorised lower_bound over the
// sorted keys gives the row offsets (empty rows included), and the column-sum kernels of
// colsums_kernels.hip do the reduction.  Deterministic: no float atomics anywhere, so
// results are bit-stable run to run and within the usual 1e-12 * sum|x| of the
// reference order.  Traffic is dominated by the sort (12 B/nnz per radix pass, read and
// written), not by the reduction; the handle API caches the sorted values, so repeated
// rowSums on a resident matrix cost one column-sum launch.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static unsigned key_bits(int32_t nrow) {
    unsigned b = 1;
    while (b < 31 && (1u << b) < (unsigned)nrow) ++b;
    return b;
}

// rocPRIM temp storage needed for the sort and the offsets search (the larger of the two)
static hipError_t rocprim_temp_bytes(int32_t nrow, int64_t nnz, size_t* bytes) {
    size_t sort_bytes = 0, search_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const double*)nullptr, (double*)nullptr, (size_t)nnz, 0u,
                                             key_bits(nrow), (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = rocprim::lower_bound(nullptr, search_bytes, (const uint32_t*)nullptr,
                             rocprim::counting_iterator<uint32_t>(0), (int32_t*)nullptr, (size_t)nnz,
                             (size_t)nrow + 1, rocprim::less<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return e;
    *bytes = sort_bytes > search_bytes ? sort_bytes : search_bytes;
    return hipSuccess;
}

hipError_t plan_row_sums(int32_t nrow, int64_t nnz, size_t colsums_ws_bytes, RowSumsLayout* L) {
    size_t temp = 0;
    hipError_t e = rocprim_temp_bytes(nrow, nnz, &temp);
    if (e != hipSuccess) return e;
    size_t off = 0;   // persistent part: the row-major form of the matrix + the reduction's carries
    L->vals_off = off;  off = align_up(off + (size_t)nnz * 8, 256);         // x sorted by row
    L->prow_off = off;  off = align_up(off + ((size_t)nrow + 1) * 4, 256);  // row offsets
    L->colsums_off = off; off = align_up(off + colsums_ws_bytes, 256);      // chunk carries
    L->persistent_bytes = off;
    off = 0;          // scratch part: only needed while building the row-major form
    L->keys_off = off;  off = align_up(off + (size_t)nnz * 4, 256);         // sorted row indices
    L->temp_off = off;  off = align_up(off + temp, 256);
    L->temp_bytes = temp;
    L->scratch_bytes = off;
    return hipSuccess;
}

// Builds the row-major value array and row offsets: `persist` receives vals + prow,
// `scratch` is free again when the stream has passed this point.
hipError_t launch_row_transpose_values(const double* d_x, const int32_t* d_i, int32_t nrow, int64_t nnz,
                                       const RowSumsLayout& L, void* persist, void* scratch,
                                       hipStream_t stream) {
    double* vals = (double*)((char*)persist + L.vals_off);
    int32_t* prow = (int32_t*)((char*)persist + L.prow_off);
    uint32_t* keys = (uint32_t*)((char*)scratch + L.keys_off);
    void* temp = (char*)scratch + L.temp_off;
    size_t temp_bytes = L.temp_bytes;
    hipError_t e = hipSuccess;
    if (nnz > 0) {
        e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)d_i, keys, d_x, vals, (size_t)nnz, 0u,
                                      key_bits(nrow), stream);
        if (e != hipSuccess) return e;
    }
    temp_bytes = L.temp_bytes;
    // prow[r] = first position whose row index is >= r  (r = 0..nrow; prow[nrow] = nnz)
    e = rocprim::lower_bound(temp, temp_bytes, (const uint32_t*)keys, rocprim::counting_iterator<uint32_t>(0),
                             prow, (size_t)nnz, (size_t)nrow + 1, rocprim::less<uint32_t>(), stream);
    return e;
}

}  // namespace rsp

// columnsums_impl.hpp -- body of the exported columnSums(), shared by the Rcpp
// build (host/rpkg/src/columnSums.cpp) and the Rcpp-free test seam (host/host_seam.cpp).
//
// Reference src/example.cpp:26-32 allocates a zero-filled NumericVector and runs
// one InnerIterator per column.  Here the allocation stays on the caller's (R
// main) thread, and the double loop becomes ONE call across the C ABI
// (include/rcppsparse_hip.h) into the HIP segmented-sum kernel.  The slots are
// handed over as raw pointers (REAL(x), INTEGER(p)); i[] is not passed: the
// reference never reads it on this path (RcppSparse.h:227 row() is not called).
// There is no CPU fallback: a non-zero status becomes a C++ exception, which
// Rcpp's END_RCPP turns into an R error (reference src/RcppExports.cpp:23).
#ifndef RCPPSPARSE_COLUMNSUMS_IMPL_HPP
#define RCPPSPARSE_COLUMNSUMS_IMPL_HPP

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/rcppsparse_hip.h"

namespace rcppsparse_core {

// Device ordinal for the one-shot path: env RCPPSPARSE_DEVICE (default 0).
inline int default_device() {
    const char* s = std::getenv("RCPPSPARSE_DEVICE");
    return s ? std::atoi(s) : 0;
}

template <class MatrixT, class Traits>
typename Traits::NumVec column_sums_via_hip(MatrixT& A) {
    const unsigned int ncol = A.cols();                       // RcppSparse.h:45
    typename Traits::NumVec sums = Traits::zeros(ncol);       // example.cpp:27
    if (ncol == 0) return sums;
    const long long nnz = (long long)A.n_nonzero();           // RcppSparse.h:48
    const double* px = nnz ? &A.x[0] : (const double*)0;
    const int rc = rsp_column_sums_host(px, &A.p[0], (int)ncol, nnz, &sums[0], default_device());
    if (rc != RSP_OK)
        throw std::runtime_error(std::string("RcppSparse columnSums (HIP): ") + rsp_last_error());
    return sums;
}

}  // namespace rcppsparse_core
#endif

// columnsums_impl.hpp -- body of the exported columnSums(), shared by the Rcpp
// build (host/rpkg/src/columnSums.cpp) and the Rcpp-free test seam (host/host_seam.cpp).
//
// Reference src/example.cpp:26-32 allocates a zero-filled NumericVector and runs
// one InnerIterator per column.  Here the allocation stays on the caller's (R
// main) thread, and the double loop becomes ONE call across the C ABI
// (include/rcppsparse_hip.h) into the HIP segmented-sum kernel.  The slots are
// handed over as raw pointers (REAL(x), INTEGER(p)); i[] is not passed: the
// reference never reads it on this path (RcppSparse.h:227 row() is not called).
//
// A machine WITHOUT any HIP device (SURVEY.md 8b "CPU fallback selected when no device", section 5): the
// reference's columnSums always answers, so this layer -- above the C ABI, never inside it -- answers too, with
// the reference's own loop (example.cpp:28-30) over THIS package's Matrix::InnerIterator into the vector that
// is already allocated.  It is selected only when rsp_device_count() reports zero devices; with a device
// present every failure stays an error (a non-zero status becomes a C++ exception, which Rcpp's END_RCPP turns
// into an R error, reference src/RcppExports.cpp:23).  RCPPSPARSE_REQUIRE_GPU=1 in the environment (or the R
// option RcppSparse.require_gpu = TRUE, which the Rcpp build passes in) switches the CPU answer off: no device
// is then the R error it was before.  This repository's tests, bench.py and smoke() all run with it set, so
// nothing measured or checked on a GPU box can come from the CPU loop; last_backend() tells which path ran.
// The C ABI itself has no fallback (librcppsparse_hip.so returns RSP_ERR_NO_DEVICE).
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

enum Backend { kBackendNone = 0, kBackendHip = 1, kBackendCpu = 2 };
inline const char* backend_name(int b) { return b == kBackendHip ? "hip" : (b == kBackendCpu ? "cpu" : "none"); }

// the path that answered the most recent columnSums() of this process (R calls in on one thread)
inline int& last_backend() {
    static int b = kBackendNone;
    return b;
}

// option: the R-level twin (1 / 0), or -1 when R has no such option set -> the environment decides
inline bool gpu_required(int option) {
    if (option >= 0) return option != 0;
    const char* s = std::getenv("RCPPSPARSE_REQUIRE_GPU");
    return s && s[0] && !(s[0] == '0' && !s[1]);
}

// the path a columnSums() call would take now: the GPU whenever one is visible; otherwise the CPU loop,
// unless a GPU is required ("none": the call is an error)
inline int choose_backend(int require_gpu_option = -1) {
    int ndev = 0;
    if (rsp_device_count(&ndev) == RSP_OK && ndev > 0) return kBackendHip;
    return gpu_required(require_gpu_option) ? kBackendNone : kBackendCpu;
}

template <class MatrixT, class Traits>
typename Traits::NumVec column_sums_via_hip(MatrixT& A, int require_gpu_option = -1) {
    const unsigned int ncol = A.cols();                       // RcppSparse.h:45
    typename Traits::NumVec sums = Traits::zeros(ncol);       // example.cpp:27
    const int backend = choose_backend(require_gpu_option);
    if (backend == kBackendCpu) {
        // no device on this machine: the reference loop, one InnerIterator per column (example.cpp:28-30)
        for (unsigned int col = 0; col < ncol; ++col)
            for (typename MatrixT::InnerIterator it(A, (int)col); it; ++it) sums[col] += it.value();
        last_backend() = kBackendCpu;
        return sums;
    }
    last_backend() = kBackendNone;
    if (ncol == 0) {
        last_backend() = backend;
        return sums;
    }
    const long long nnz = (long long)A.n_nonzero();           // RcppSparse.h:48
    const double* px = nnz ? &A.x[0] : (const double*)0;
    // (no device and a GPU required: the shim's RSP_ERR_NO_DEVICE is the error)
    const int rc = rsp_column_sums_host(px, &A.p[0], (int)ncol, nnz, &sums[0], default_device());
    if (rc != RSP_OK)
        throw std::runtime_error(std::string("RcppSparse columnSums (HIP): ") + rsp_last_error());
    last_backend() = kBackendHip;
    return sums;
}

}  // namespace rcppsparse_core
#endif

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
// WHICH PATH ANSWERS (SURVEY.md 8b "CPU fallback selected when no device", section 5 "min-nnz threshold for GPU
// offload").  The reference's columnSums always answers and costs a few nanoseconds per entry; a call through the
// device costs two transfers, two launches and a synchronisation whatever the size (~0.1 ms), so on a SMALL matrix
// the reference's own loop on the host is the faster answer and the drop-in must never be slower than what it
// replaces.  This layer -- above the C ABI, never inside it -- therefore answers with the reference's loop
// (example.cpp:28-30) over THIS package's Matrix::InnerIterator
//   * when rsp_device_count() reports zero devices, and
//   * when the matrix holds fewer than min_nnz stored entries: options(RcppSparse.min_nnz = n) in R, else
//     RCPPSPARSE_MIN_NNZ in the environment, else kDefaultMinNnz -- the crossover measured on an MI355X box
//     (tools/measure_host_path.py, profiles/r05_one_shot.json); 0 sends every matrix to the GPU.
// With a device present and the matrix large enough every failure stays an error (a non-zero status becomes a C++
// exception, which Rcpp's END_RCPP turns into an R error, reference src/RcppExports.cpp:23).
// RCPPSPARSE_REQUIRE_GPU=1 in the environment (or the R option RcppSparse.require_gpu = TRUE, which the Rcpp build
// passes in) switches BOTH host answers off: every call goes to the device, and no device is the R error it was
// before.  This repository's tests, bench.py and smoke() all run with it set, so nothing measured or checked on a
// GPU box can come from the CPU loop; last_backend() tells which path ran.
// The C ABI itself has no fallback (librcppsparse_hip.so returns RSP_ERR_NO_DEVICE) and no threshold.
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

// Several GPUs for the one-shot path (opt-in): RCPPSPARSE_DEVICES = "all" (one column range per visible device) or a
// comma-separated list of ordinals ("0,1,2,3"; an ordinal may repeat).  The one-shot call is bound by the host link
// (C3 145 ms at 55 GB/s against 1.2 ms of kernel); rsp_column_sums_host_multi cuts the columns into
// nnz-balanced ranges and sends every range over ITS device's link from a host thread of its own.  Not set (default): one
// device, RCPPSPARSE_DEVICE.  Returns the number of listed devices (0: not set, -1: "all"); ordinals into out[0..n).
inline int devices_setting(int* out, int capacity) {
    const char* s = std::getenv("RCPPSPARSE_DEVICES");
    if (!s || !s[0]) return 0;
    if (s[0] == 'a' || s[0] == 'A') return -1;
    int n = 0;
    while (*s && n < capacity) {
        char* end = 0;
        const long v = std::strtol(s, &end, 10);
        if (end == s) break;
        out[n++] = (int)v;
        s = end;
        while (*s == ',' || *s == ' ') ++s;
    }
    return n;
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

// Stored entries below which the host loop answers although a device is present.  Measured on an MI355X box
// (profiles/r05_one_shot.json): the one-shot device call and the 1-thread loop cross between 1e5 and 1e6 entries.
constexpr long long kDefaultMinNnz = 250000;

// option: the R-level twin (options(RcppSparse.min_nnz = n)), or -1 when R has no such option -> environment, default
inline long long min_nnz_setting(long long option = -1) {
    if (option >= 0) return option;
    const char* s = std::getenv("RCPPSPARSE_MIN_NNZ");
    if (s && s[0]) {
        const long long v = std::atoll(s);
        return v < 0 ? 0 : v;
    }
    return kDefaultMinNnz;
}

// the path a columnSums() call on a matrix of `nnz` stored entries would take now (nnz < 0: a matrix large enough
// for the device): the GPU when one is visible and the matrix is not below the threshold; otherwise the CPU loop;
// a GPU required: always the GPU ("none" without one: the call is an error)
inline int choose_backend(int require_gpu_option = -1, long long nnz = -1, long long min_nnz_option = -1) {
    int ndev = 0;
    const bool have = rsp_device_count(&ndev) == RSP_OK && ndev > 0;
    if (gpu_required(require_gpu_option)) return have ? kBackendHip : kBackendNone;
    if (!have) return kBackendCpu;
    return (nnz >= 0 && nnz < min_nnz_setting(min_nnz_option)) ? kBackendCpu : kBackendHip;
}

template <class MatrixT, class Traits>
typename Traits::NumVec column_sums_via_hip(MatrixT& A, int require_gpu_option = -1, long long min_nnz_option = -1) {
    const unsigned int ncol = A.cols();                       // RcppSparse.h:45
    typename Traits::NumVec sums = Traits::zeros(ncol);       // example.cpp:27
    const int backend = choose_backend(require_gpu_option, (long long)A.n_nonzero(), min_nnz_option);
    if (backend == kBackendCpu) {
        // no device on this machine, or a matrix below the offload threshold: the reference loop, one InnerIterator
        // per column (example.cpp:28-30)
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
    int devs[64];
    const int ndev = devices_setting(devs, 64);
    const int rc = ndev == 0 ? rsp_column_sums_host(px, &A.p[0], (int)ncol, nnz, &sums[0], default_device())
                             : rsp_column_sums_host_multi(px, &A.p[0], (int)ncol, nnz, &sums[0], ndev < 0 ? (const int*)0 : devs,
                                                          ndev < 0 ? 0 : ndev);
    if (rc != RSP_OK)
        throw std::runtime_error(std::string("RcppSparse columnSums (HIP): ") + rsp_last_error());
    last_backend() = kBackendHip;
    return sums;
}

}  // namespace rcppsparse_core
#endif

#include "../inst/include/RcppSparse.h"
#include "../inst/include/columnsums_impl.hpp"

namespace {
// options(RcppSparse.require_gpu = TRUE / FALSE): 1 / 0; not set: -1 (the environment variable decides)
int require_gpu_option() {
    SEXP o = Rf_GetOption1(Rf_install("RcppSparse.require_gpu"));
    if (o == R_NilValue) return -1;
    return Rf_asLogical(o) == TRUE ? 1 : 0;
}
// options(RcppSparse.min_nnz = n): matrices with fewer stored entries are summed by the host loop; not set: -1
// (RCPPSPARSE_MIN_NNZ in the environment, else the measured default)
long long min_nnz_option() {
    SEXP o = Rf_GetOption1(Rf_install("RcppSparse.min_nnz"));
    if (o == R_NilValue) return -1;
    const double v = Rf_asReal(o);
    return v >= 0 ? (long long)v : -1;
}
}  // namespace

//' Sum every column of a sparse matrix on the GPU
//'
//' Takes a \code{dgCMatrix} and returns a plain numeric vector with one sum per column,
//' like the CPU original.  Instead of walking each column with a
//' \code{RcppSparse::Matrix::InnerIterator}, the slots \code{x} and \code{p} are handed to a
//' HIP segmented-sum kernel (AMD Instinct MI355X) through the C interface in
//' \code{rcppsparse_hip.h}.  On a machine without any GPU the function still answers, like the CPU
//' original: the same column loop runs on the host (\code{columnSumsBackend(last = TRUE)} then says
//' \code{"cpu"}).  The host loop also answers for small matrices -- fewer than
//' \code{getOption("RcppSparse.min_nnz")} stored entries (else \code{RCPPSPARSE_MIN_NNZ} in the environment,
//' else 250000, the measured crossover): a trip through the GPU costs about 0.1 ms whatever the size, the loop
//' a few nanoseconds per entry, and the function is never slower than the CPU original.
//' On a machine with several GPUs \code{RCPPSPARSE_DEVICES=all} (or a list such as \code{0,1,2,3}) in the
//' environment spreads the call over them: the matrix is cut into column ranges of equal numbers of stored entries
//' and every range travels over its own GPU's host link, which is what bounds a call on host data.
//' \code{options(RcppSparse.require_gpu = TRUE)} or \code{RCPPSPARSE_REQUIRE_GPU=1} in the
//' environment turn both off: every call goes to the GPU and no GPU is then an R error.  With a GPU present
//' a failure is always an error.
//'
//' @param A a \code{dgCMatrix} (package Matrix)
//' @return numeric vector of length \code{ncol(A)}
//' @examples
//' \dontrun{
//' A <- Matrix::rsparsematrix(1e5, 1e3, density = 0.1)
//' stopifnot(all.equal(columnSums(A), Matrix::colSums(A)))
//' }
//[[Rcpp::export]]
Rcpp::NumericVector columnSums(RcppSparse::Matrix& A) {
    // the result vector is allocated here, on the R main thread; the shim only fills it
    return rcppsparse_core::column_sums_via_hip<RcppSparse::Matrix, RcppSparse::RcppTraits>(A, require_gpu_option(), min_nnz_option());
}

// (no R-level export: called by R_unload_RcppSparse in rcpp_glue.cpp)
void releaseCached() { (void)rsp_release_cached(); }

//' Which path answers columnSums()
//'
//' @param last \code{FALSE}: the path a call of \code{columnSums()} on a \code{dgCMatrix} would take now;
//'   \code{TRUE}: the path that answered the most recent call
//' @return \code{"hip"} (the GPU), \code{"cpu"} (no GPU on this machine: the host loop) or \code{"none"}
//'   (no GPU and a GPU is required: \code{columnSums()} is an error; or no call yet)
//[[Rcpp::export]]
SEXP columnSumsBackend(int last) {
    const int b = last ? rcppsparse_core::last_backend() : rcppsparse_core::choose_backend(require_gpu_option());
    return Rcpp::wrap(rcppsparse_core::backend_name(b));
}

#include "../inst/include/RcppSparse.h"
#include "../inst/include/columnsums_impl.hpp"

//' Sum every column of a sparse matrix on the GPU
//'
//' Takes a \code{dgCMatrix} and returns a plain numeric vector with one sum per column,
//' like the CPU original.  Instead of walking each column with a
//' \code{RcppSparse::Matrix::InnerIterator}, the slots \code{x} and \code{p} are handed to a
//' HIP segmented-sum kernel (AMD Instinct MI355X) through the C interface in
//' \code{rcppsparse_hip.h}.  A machine without a usable GPU gets an R error, never a
//' silently different code path.
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
    return rcppsparse_core::column_sums_via_hip<RcppSparse::Matrix, RcppSparse::RcppTraits>(A);
}

#include "../inst/include/RcppSparse.h"
#include "../inst/include/columnsums_impl.hpp"

//' Column sums of a sparse matrix on the GPU
//'
//' Same contract as the CPU original: takes a \code{dgCMatrix}, returns a plain
//' numeric vector of length \code{ncol(A)}.  The per-column
//' \code{RcppSparse::Matrix::InnerIterator} loop is replaced by one call into a
//' HIP segmented-sum kernel (MI355X / gfx950) through the C interface declared
//' in \code{rcppsparse_hip.h}; there is no CPU fallback, a missing GPU is an R error.
//'
//' @param A an object of class \code{dgCMatrix}
//' @examples
//' library(Matrix)
//' A <- rsparsematrix(nrow = 10, ncol = 5, density = 0.5)
//' columnSums(A)
//[[Rcpp::export]]
Rcpp::NumericVector columnSums(RcppSparse::Matrix& A) {
    // allocation of the result happens here, on the R main thread; the shim only fills it
    return rcppsparse_core::column_sums_via_hip<RcppSparse::Matrix, RcppSparse::RcppTraits>(A);
}

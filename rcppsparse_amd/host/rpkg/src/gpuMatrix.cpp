#include "../inst/include/RcppSparse.h"
#include "../inst/include/rcppsparse_hip.h"

#include <cstdlib>
#include <stdexcept>
#include <string>

// GPU-resident dgCMatrix behind an R external pointer ("next" row f2 of SURVEY.md section 8).
//
// The slots x / i / p / Dim (reference inst/include/RcppSparse.h:29-41, the layout wrap() writes
// back at :387-394) are the wire format: gpuMatrix(A) uploads them once and hands R an external
// pointer that owns the DEVICE copy only; columnSums() on that handle runs the same kernels on
// the resident copy and pays no PCIe transfer of x.  The handle is explicit on purpose: caching
// device copies by SEXP address would be wrong, because R objects are mutable in place through
// this very class (reference vignettes/Documentation.Rmd:335-347) and addresses are reused after
// a garbage collection.  A later change of A is therefore NOT seen by an existing handle.
//
// Lifetime: the external pointer carries a C finalizer that calls rsp_csc_free when R collects
// it (or at gpuFree(), whichever comes first; a released handle raises an R error afterwards).

namespace {

void release_gpu_matrix(rsp_csc* h) { rsp_csc_free(h); }   // rsp_csc_free(NULL) is fine

typedef Rcpp::XPtr<rsp_csc, Rcpp::PreserveStorage, release_gpu_matrix, true> GpuMatrixPtr;

// The two kinds of handle are told apart by the external pointer's TAG (a symbol set here, which R code cannot
// change), not by the class attribute (which it can): a "gpuMatrixMulti" reaching a single-GPU routine, or the
// reverse -- through class<- or a direct .Call -- is an R error instead of one struct read as the other.
SEXP single_tag() { return Rf_install("rsp_csc"); }
SEXP multi_tag() { return Rf_install("rsp_mcsc"); }

void check_tag(SEXP handle, SEXP want, const char* what) {
    if (R_ExternalPtrTag(handle) != want) throw std::invalid_argument(std::string("not a ") + what + " handle");
}

rsp_csc* resident(SEXP handle) {
    GpuMatrixPtr ptr(handle);
    check_tag(handle, single_tag(), "gpuMatrix");
    rsp_csc* h = ptr.get();
    if (!h) throw std::invalid_argument("this gpuMatrix handle has been released");
    return h;
}

// Dim[0] / Dim[1] come from the native handle, not from the (mutable) "Dim" attribute: an edited attribute must
// not decide how long an output vector is
void native_dims(rsp_csc* h, int* nrow, int* ncol) {
    int32_t r = 0, c = 0;
    if (rsp_csc_dims(h, &r, &c, 0) != RSP_OK) throw std::runtime_error(rsp_last_error());
    *nrow = r;
    *ncol = c;
}

}  // namespace

//' Keep a sparse matrix in GPU memory
//'
//' Uploads the slots of a \code{dgCMatrix} to one GPU and returns a handle (an external
//' pointer of class \code{"gpuMatrix"}).  \code{columnSums()} on the handle does not transfer
//' the matrix again.  The handle holds a copy: later changes of \code{A} are not seen.
//'
//' @param A a \code{dgCMatrix}
//' @param device GPU ordinal (default: environment variable \code{RCPPSPARSE_DEVICE}, else 0)
//' @return external pointer of class \code{"gpuMatrix"} with attribute \code{Dim}
//[[Rcpp::export]]
SEXP gpuMatrix(RcppSparse::Matrix& A, int device) {
    const long long nnz = (long long)A.n_nonzero();
    rsp_csc_t h = 0;
    const int rc = rsp_csc_upload(nnz ? &A.x[0] : (const double*)0, nnz ? &A.i[0] : (const int*)0, &A.p[0],
                                  (int)A.rows(), (int)A.cols(), nnz, device, &h);
    if (rc != RSP_OK) throw std::runtime_error(std::string("RcppSparse gpuMatrix (HIP): ") + rsp_last_error());
    GpuMatrixPtr ptr(h, true, single_tag());        // R now owns the device copy
    ptr.attr("class") = "gpuMatrix";
    ptr.attr("Dim") = Rcpp::IntegerVector::create((int)A.rows(), (int)A.cols());
    return ptr;
}

//' Column sums of a GPU-resident matrix
//' @param handle a \code{"gpuMatrix"}
//' @return numeric vector of length \code{ncol}
//[[Rcpp::export]]
Rcpp::NumericVector gpuColumnSums(SEXP handle) {
    rsp_csc* h = resident(handle);
    int nrow = 0, ncol = 0;
    native_dims(h, &nrow, &ncol);
    Rcpp::NumericVector sums(ncol);                  // allocated by R, on the R main thread
    if (ncol == 0) return sums;
    if (rsp_csc_column_sums(h, &sums[0]) != RSP_OK)
        throw std::runtime_error(std::string("RcppSparse columnSums (HIP): ") + rsp_last_error());
    return sums;
}

//' Column means, row sums and row means of a GPU-resident matrix
//'
//' The device forms of \code{Matrix::colSums()}, \code{colMeans()}, \code{rowSums()} and \code{rowMeans()}
//' of the C++ class (reference inst/include/RcppSparse.h:131-156) on the resident copy.
//' @param handle a \code{"gpuMatrix"}
//' @param what 0 column sums, 1 column means, 2 row sums, 3 row means
//' @return numeric vector of length \code{ncol} (0, 1) or \code{nrow} (2, 3)
//[[Rcpp::export]]
Rcpp::NumericVector gpuReduce(SEXP handle, int what) {
    rsp_csc* h = resident(handle);
    if (what < 0 || what > 3) throw std::invalid_argument("what must be 0 (colSums), 1 (colMeans), 2 (rowSums) or 3 (rowMeans)");
    int nrow = 0, ncol = 0;
    native_dims(h, &nrow, &ncol);
    const int n = what < 2 ? ncol : nrow;
    Rcpp::NumericVector out(n);                      // allocated by R, on the R main thread
    if (n == 0) return out;
    const int rc = what == 0 ? rsp_csc_column_sums(h, &out[0]) : what == 1 ? rsp_csc_column_means(h, &out[0])
                 : what == 2 ? rsp_csc_row_sums(h, &out[0]) : rsp_csc_row_means(h, &out[0]);
    if (rc != RSP_OK) throw std::runtime_error(std::string("RcppSparse gpuMatrix (HIP): ") + rsp_last_error());
    return out;
}

//' t(A) %*% A of a GPU-resident matrix
//'
//' The device form of \code{Matrix::crossprod()} (reference inst/include/RcppSparse.h:159-194): dense
//' \code{ncol x ncol}.
//' @param handle a \code{"gpuMatrix"}
//' @return numeric matrix
//[[Rcpp::export]]
Rcpp::NumericMatrix gpuCrossprod(SEXP handle) {
    rsp_csc* h = resident(handle);
    int nrow = 0, n = 0;
    native_dims(h, &nrow, &n);
    Rcpp::NumericMatrix out(n, n);
    if (n == 0) return out;
    if (rsp_csc_crossprod(h, &out(0, 0)) != RSP_OK)
        throw std::runtime_error(std::string("RcppSparse crossprod (HIP): ") + rsp_last_error());
    return out;
}

// ---- the same handle spread over several GPUs of the node (rsp_mcsc_*) ---------------------------------
namespace {

void release_gpu_matrix_multi(rsp_mcsc* h) { rsp_mcsc_free(h); }

typedef Rcpp::XPtr<rsp_mcsc, Rcpp::PreserveStorage, release_gpu_matrix_multi, true> GpuMatrixMultiPtr;

}  // namespace

//' Keep a sparse matrix in the memory of several GPUs
//'
//' The columns are cut into nnz-balanced contiguous ranges, one per entry of \code{devices} (an ordinal may
//' repeat); every range is uploaded over its own GPU's host link and stays resident there.
//' @param A a \code{dgCMatrix}
//' @param devices GPU ordinals, one per shard
//' @return external pointer of class \code{"gpuMatrixMulti"} with attribute \code{Dim}
//[[Rcpp::export]]
SEXP gpuMatrixMulti(RcppSparse::Matrix& A, Rcpp::IntegerVector devices) {
    const long long nnz = (long long)A.n_nonzero();
    if (devices.size() < 1) throw std::invalid_argument("devices must name at least one GPU");
    rsp_mcsc_t h = 0;
    const int rc = rsp_mcsc_upload_csc(nnz ? &A.x[0] : (const double*)0, nnz ? &A.i[0] : (const int*)0, &A.p[0],
                                       (int)A.rows(), (int)A.cols(), nnz, &devices[0], (int)devices.size(), &h);
    if (rc != RSP_OK) throw std::runtime_error(std::string("RcppSparse gpuMatrix (HIP): ") + rsp_last_error());
    GpuMatrixMultiPtr ptr(h, true, multi_tag());
    ptr.attr("class") = "gpuMatrixMulti";
    ptr.attr("Dim") = Rcpp::IntegerVector::create((int)A.rows(), (int)A.cols());
    return ptr;
}

//' Column / row sums and means of a matrix resident on several GPUs
//' @param handle a \code{"gpuMatrixMulti"}
//' @param what 0 column sums, 1 column means, 2 row sums, 3 row means
//[[Rcpp::export]]
Rcpp::NumericVector gpuMultiReduce(SEXP handle, int what) {
    GpuMatrixMultiPtr ptr(handle);
    check_tag(handle, multi_tag(), "gpuMatrixMulti");
    rsp_mcsc* h = ptr.get();
    if (!h) throw std::invalid_argument("this gpuMatrix handle has been released");
    if (what < 0 || what > 3) throw std::invalid_argument("what must be 0 (colSums), 1 (colMeans), 2 (rowSums) or 3 (rowMeans)");
    int32_t nrow = 0, ncol = 0;
    if (rsp_mcsc_dims(h, &nrow, &ncol, 0) != RSP_OK) throw std::runtime_error(rsp_last_error());
    const int n = what < 2 ? ncol : nrow;
    Rcpp::NumericVector out(n);
    if (n == 0) return out;
    const int rc = what == 0 ? rsp_mcsc_column_sums(h, &out[0]) : what == 1 ? rsp_mcsc_column_means(h, &out[0])
                 : what == 2 ? rsp_mcsc_row_sums(h, &out[0]) : rsp_mcsc_row_means(h, &out[0]);
    if (rc != RSP_OK) throw std::runtime_error(std::string("RcppSparse gpuMatrix (HIP): ") + rsp_last_error());
    return out;
}

//' @rdname gpuMatrixMulti
//' @param handle a \code{"gpuMatrixMulti"}
//[[Rcpp::export]]
void gpuFreeMulti(SEXP handle) {
    GpuMatrixMultiPtr ptr(handle);
    check_tag(handle, multi_tag(), "gpuMatrixMulti");
    ptr.release();
}

//' Release the GPU copy now instead of at garbage collection
//' @param handle a \code{"gpuMatrix"}
//[[Rcpp::export]]
void gpuFree(SEXP handle) {
    GpuMatrixPtr ptr(handle);
    check_tag(handle, single_tag(), "gpuMatrix");
    ptr.release();                                   // runs the finalizer once and clears the pointer
}

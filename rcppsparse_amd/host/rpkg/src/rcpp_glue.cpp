// rcpp_glue.cpp -- hand-written R <-> C++ registration of the package's native routines.
//
// R reaches native code through C symbols of the package's shared object: the routines
// themselves and the init hook R runs right after dlopen().  A drop-in for the reference package
// has to export the reference's two symbols under the reference's names (reference
// src/RcppExports.cpp:16 and :31) and register `_RcppSparse_columnSums` with arity 1 and dynamic
// symbol lookup switched off, because the R wrapper calls it by its registered name
// (`.Call(`_RcppSparse_columnSums`, A)`).  The gpuMatrix routines after it are additions
// of this package.
//
// Rcpp::compileAttributes() can regenerate an equivalent file from the [[Rcpp::export]]
// attributes in columnSums.cpp / gpuMatrix.cpp; delete this one if you do.  Nothing here touches
// HIP: loading the package must stay cheap and must succeed on a machine without a GPU (the
// device is first touched inside columnSums() / gpuMatrix(); without one columnSums() answers on the
// host, see columnsums_impl.hpp).
#include "../inst/include/RcppSparse.h"
#include <Rcpp.h>

// A package built with -DRCPP_USE_GLOBAL_ROSTREAM must define the two stream objects itself
// (reference src/RcppExports.cpp:9-12); without these lines such a build fails to link.
#ifdef RCPP_USE_GLOBAL_ROSTREAM
Rcpp::Rostream<true>& Rcpp::Rcout = Rcpp::Rcpp_cout_get();
Rcpp::Rostream<false>& Rcpp::Rcerr = Rcpp::Rcpp_cerr_get();
#endif

// the exported C++ functions (columnSums.cpp, gpuMatrix.cpp)
Rcpp::NumericVector columnSums(RcppSparse::Matrix& A);
SEXP gpuMatrix(RcppSparse::Matrix& A, int device);
Rcpp::NumericVector gpuColumnSums(SEXP handle);
void gpuFree(SEXP handle);
Rcpp::NumericVector gpuReduce(SEXP handle, int what);
Rcpp::NumericMatrix gpuCrossprod(SEXP handle);
SEXP gpuMatrixMulti(RcppSparse::Matrix& A, Rcpp::IntegerVector devices);
Rcpp::NumericVector gpuMultiReduce(SEXP handle, int what);
void gpuFreeMulti(SEXP handle);
SEXP columnSumsBackend(int last);
void releaseCached();   // columnSums.cpp: rsp_release_cached() of the C ABI

namespace {

// Every .Call entry point has the shape of reference src/RcppExports.cpp:16-24:
// BEGIN_RCPP / END_RCPP turn any C++ exception -- a missing dgCMatrix slot, a HIP failure
// reported by the shim -- into an R condition instead of unwinding through R's C stack; the
// RNGScope keeps R's random-number state consistent around the call exactly like the
// generated glue does (:19; nothing here draws random numbers).
SEXP call_columnSums(SEXP dgCMatrix) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    // Exporter<RcppSparse::Matrix> wraps the four slots by reference; `A` keeps them alive
    // (protected from the garbage collector) until the end of this call
    Rcpp::traits::input_parameter<RcppSparse::Matrix&>::type A(dgCMatrix);
    result = Rcpp::wrap(columnSums(A));
    return result;
    END_RCPP
}

SEXP call_gpuMatrix(SEXP dgCMatrix, SEXP device) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    Rcpp::traits::input_parameter<RcppSparse::Matrix&>::type A(dgCMatrix);
    result = gpuMatrix(A, Rcpp::as<int>(device));
    return result;
    END_RCPP
}

SEXP call_gpuColumnSums(SEXP handle) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    result = Rcpp::wrap(gpuColumnSums(handle));
    return result;
    END_RCPP
}

SEXP call_gpuFree(SEXP handle) {
    BEGIN_RCPP
    Rcpp::RNGScope rng_state;
    gpuFree(handle);
    return R_NilValue;
    END_RCPP
}

SEXP call_gpuReduce(SEXP handle, SEXP what) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    result = Rcpp::wrap(gpuReduce(handle, Rcpp::as<int>(what)));
    return result;
    END_RCPP
}

SEXP call_gpuCrossprod(SEXP handle) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    result = Rcpp::wrap(gpuCrossprod(handle));
    return result;
    END_RCPP
}

SEXP call_gpuMatrixMulti(SEXP dgCMatrix, SEXP devices) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    Rcpp::traits::input_parameter<RcppSparse::Matrix&>::type A(dgCMatrix);
    result = gpuMatrixMulti(A, Rcpp::IntegerVector(devices));
    return result;
    END_RCPP
}

SEXP call_gpuMultiReduce(SEXP handle, SEXP what) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    result = Rcpp::wrap(gpuMultiReduce(handle, Rcpp::as<int>(what)));
    return result;
    END_RCPP
}

SEXP call_gpuFreeMulti(SEXP handle) {
    BEGIN_RCPP
    Rcpp::RNGScope rng_state;
    gpuFreeMulti(handle);
    return R_NilValue;
    END_RCPP
}

SEXP call_columnSumsBackend(SEXP last) {
    BEGIN_RCPP
    Rcpp::RObject result;
    Rcpp::RNGScope rng_state;
    result = columnSumsBackend(Rcpp::as<int>(last));
    return result;
    END_RCPP
}

}  // namespace

extern "C" {

SEXP _RcppSparse_columnSums(SEXP A) { return call_columnSums(A); }
SEXP _RcppSparse_gpuMatrix(SEXP A, SEXP device) { return call_gpuMatrix(A, device); }
SEXP _RcppSparse_gpuColumnSums(SEXP handle) { return call_gpuColumnSums(handle); }
SEXP _RcppSparse_gpuFree(SEXP handle) { return call_gpuFree(handle); }
SEXP _RcppSparse_gpuReduce(SEXP handle, SEXP what) { return call_gpuReduce(handle, what); }
SEXP _RcppSparse_gpuCrossprod(SEXP handle) { return call_gpuCrossprod(handle); }
SEXP _RcppSparse_gpuMatrixMulti(SEXP A, SEXP devices) { return call_gpuMatrixMulti(A, devices); }
SEXP _RcppSparse_gpuMultiReduce(SEXP handle, SEXP what) { return call_gpuMultiReduce(handle, what); }
SEXP _RcppSparse_gpuFreeMulti(SEXP handle) { return call_gpuFreeMulti(handle); }
SEXP _RcppSparse_columnSumsBackend(SEXP last) { return call_columnSumsBackend(last); }

void R_init_RcppSparse(DllInfo* dll) {
    static const R_CallMethodDef routines[] = {
        {"_RcppSparse_columnSums", reinterpret_cast<DL_FUNC>(&_RcppSparse_columnSums), 1},   // the reference's
        {"_RcppSparse_gpuMatrix", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuMatrix), 2},
        {"_RcppSparse_gpuColumnSums", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuColumnSums), 1},
        {"_RcppSparse_gpuFree", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuFree), 1},
        {"_RcppSparse_gpuReduce", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuReduce), 2},
        {"_RcppSparse_gpuCrossprod", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuCrossprod), 1},
        {"_RcppSparse_gpuMatrixMulti", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuMatrixMulti), 2},
        {"_RcppSparse_gpuMultiReduce", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuMultiReduce), 2},
        {"_RcppSparse_gpuFreeMulti", reinterpret_cast<DL_FUNC>(&_RcppSparse_gpuFreeMulti), 1},
        {"_RcppSparse_columnSumsBackend", reinterpret_cast<DL_FUNC>(&_RcppSparse_columnSumsBackend), 1},
        {NULL, NULL, 0}};
    R_registerRoutines(dll, /*.C*/ NULL, /*.Call*/ routines, /*.Fortran*/ NULL, /*.External*/ NULL);
    R_useDynamicSymbols(dll, FALSE);   // only registered names resolve
}

// R runs this when the package's shared object is unloaded (library.dynam.unload / detach(unload = TRUE)): the C library
// keeps a stream and a few device buffers per GPU between columnSums() calls (rsp_column_sums_host) -- give them back.
void R_unload_RcppSparse(DllInfo*) { releaseCached(); }

}  // extern "C"

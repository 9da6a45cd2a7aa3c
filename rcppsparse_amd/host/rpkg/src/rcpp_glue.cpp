// rcpp_glue.cpp -- R <-> C++ registration for the one exported routine.
//
// Stands in for the file Rcpp::compileAttributes() generates in the reference
// (src/RcppExports.cpp): the shared object must export exactly
//   SEXP _RcppSparse_columnSums(SEXP)   and   void R_init_RcppSparse(DllInfo*)
// and register the routine with arity 1, dynamic lookup off.  Re-running
// compileAttributes() on src/example.cpp regenerates an equivalent file; delete
// this one if you do.  Loading the package must stay cheap and must succeed on a
// machine without a GPU, so nothing here touches HIP.
#include "../inst/include/RcppSparse.h"
#include <Rcpp.h>

Rcpp::NumericVector columnSums(RcppSparse::Matrix& A);

extern "C" SEXP _RcppSparse_columnSums(SEXP A_sexp) {
    BEGIN_RCPP                                   // C++ exceptions -> R conditions
    Rcpp::traits::input_parameter<RcppSparse::Matrix&>::type A(A_sexp);   // S4 -> Matrix, zero-copy
    Rcpp::RObject result = Rcpp::wrap(columnSums(A));
    return result;
    END_RCPP
}

static const R_CallMethodDef call_entries[] = {
    {"_RcppSparse_columnSums", (DL_FUNC)&_RcppSparse_columnSums, 1},
    {NULL, NULL, 0}};

extern "C" void R_init_RcppSparse(DllInfo* dll) {
    R_registerRoutines(dll, NULL, call_entries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

// rcpp_glue.cpp -- hand-written R <-> C++ registration for the package's single routine.
//
// R reaches native code through two C symbols of the package's shared object: the routine
// itself and the init hook R runs right after dlopen().  A drop-in for the reference package
// has to export them under the reference's names (reference src/RcppExports.cpp:16 and :31)
// and register the routine with arity 1 and dynamic symbol lookup switched off, because the
// R wrapper calls it by its registered name (`.Call(`_RcppSparse_columnSums`, A)`).
//
// Rcpp::compileAttributes() can regenerate an equivalent file from the [[Rcpp::export]]
// attribute in columnSums.cpp; delete this one if you do.  Nothing here touches HIP: loading
// the package must stay cheap and must succeed on a machine without a GPU (the device is
// first touched inside columnSums()).
#include "../inst/include/RcppSparse.h"
#include <Rcpp.h>

// the exported C++ function, defined in columnSums.cpp
Rcpp::NumericVector columnSums(RcppSparse::Matrix& A);

namespace {

// One .Call entry point: SEXP in (an S4 dgCMatrix), SEXP out (a numeric vector).
// BEGIN_RCPP / END_RCPP turn any C++ exception -- a missing dgCMatrix slot, a HIP failure
// reported by the shim -- into an R condition instead of unwinding through R's C stack.
SEXP call_columnSums(SEXP dgCMatrix) {
    BEGIN_RCPP
    // Exporter<RcppSparse::Matrix> wraps the four slots by reference; `A` keeps them alive
    // (protected from the garbage collector) until the end of this call
    Rcpp::traits::input_parameter<RcppSparse::Matrix&>::type A(dgCMatrix);
    Rcpp::NumericVector sums = columnSums(A);
    return Rcpp::wrap(sums);
    END_RCPP
}

}  // namespace

extern "C" {

SEXP _RcppSparse_columnSums(SEXP A) { return call_columnSums(A); }

void R_init_RcppSparse(DllInfo* dll) {
    static const R_CallMethodDef routines[] = {
        {"_RcppSparse_columnSums", reinterpret_cast<DL_FUNC>(&_RcppSparse_columnSums), 1},
        {NULL, NULL, 0}};
    R_registerRoutines(dll, /*.C*/ NULL, /*.Call*/ routines, /*.Fortran*/ NULL, /*.External*/ NULL);
    R_useDynamicSymbols(dll, FALSE);   // only registered names resolve
}

}  // extern "C"

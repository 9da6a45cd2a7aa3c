// TEST SCAFFOLDING -- an API-shaped stand-in for the parts of Rcpp that THIS repository's
// Rcpp layer (rcppsparse_amd/host/RcppSparse.h, rpkg/src/*.cpp) touches, so that layer can
// at least be compiled, linked and driven in an image that has neither R nor Rcpp.  It is
// not Rcpp, mimics only what is used, and is never part of the product or of any reference
// build (the reference is not compiled anywhere in this repository).
#ifndef MOCK_RCPPCOMMON_H
#define MOCK_RCPPCOMMON_H
#include <cstddef>
struct SEXPREC;
typedef SEXPREC* SEXP;
typedef std::ptrdiff_t R_xlen_t;
namespace Rcpp {
namespace traits {
template <class T>
class Exporter;   // primary template: specialised by headers that extend Rcpp::as<>
}
}  // namespace Rcpp
#endif

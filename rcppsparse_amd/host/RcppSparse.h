// RcppSparse.h -- drop-in replacement for the reference's inst/include/RcppSparse.h.
//
// Same include contract as the reference (its lines 1-18; README.md:15): include
// this header BEFORE Rcpp.h, because Rcpp::as<RcppSparse::Matrix>() is enabled by
// specialising Rcpp::traits::Exporter between RcppCommon.h and Rcpp.h.
//
// What is different from the reference header: the class body lives in the
// storage-agnostic template rcppsparse_core::CscMatrix (rcppsparse_core.hpp, no
// R needed, unit-tested through host/host_seam.cpp); this file only binds it to
// Rcpp's vector types and adds the three members that need R objects (clone,
// wrap, the S4 constructor) plus a native transpose().  Pure C++/Rcpp: no HIP
// include, so `sourceCpp` consumers build on machines without ROCm.  The GPU path
// is behind the package's compiled, exported columnSums() (rpkg/src/columnSums.cpp).
//
// Needs R + Rcpp (>= 1.0.7) to compile; neither is in the build image of this
// repository, so this file is checked by review and by compiling the shared
// template against the seam's stand-in types.
#ifndef RCPPSPARSE_H
#define RCPPSPARSE_H

#include <RcppCommon.h>

namespace RcppSparse {
class Matrix;
}

namespace Rcpp {
namespace traits {
template <>
class Exporter<RcppSparse::Matrix>;
}
}  // namespace Rcpp

#include <Rcpp.h>

//[[Rcpp::plugins(openmp)]]
#ifdef _OPENMP
#include <omp.h>
#endif

#include <algorithm>
#include <string>
#include <vector>

#include "rcppsparse_core.hpp"

namespace RcppSparse {

// How the core template allocates and reads Rcpp objects.
struct RcppTraits {
    typedef Rcpp::NumericVector NumVec;
    typedef Rcpp::IntegerVector IntVec;
    typedef Rcpp::NumericMatrix NumMat;
    static NumVec zeros(R_xlen_t n) { return NumVec(n); }                   // zero-filled by Rcpp
    static NumMat zeros(int nr, int nc) { return NumMat(nr, nc); }
    static NumVec num_slot(const Rcpp::S4& s, const char* name) { return s.slot(name); }
    static IntVec int_slot(const Rcpp::S4& s, const char* name) { return s.slot(name); }
};

class Matrix : public rcppsparse_core::CscMatrix<RcppTraits> {
    typedef rcppsparse_core::CscMatrix<RcppTraits> Base;

public:
    // constructors (reference RcppSparse.h:33-42): zero-copy, by reference
    Matrix() {}
    Matrix(Rcpp::NumericVector x, Rcpp::IntegerVector i, Rcpp::IntegerVector p, Rcpp::IntegerVector Dim)
        : Base(x, i, p, Dim) {}
    Matrix(const Rcpp::S4& s) { assign_from_slots(s); }   // throws std::invalid_argument if a slot is missing

    // deep copy of the four R vectors (reference :54-60)
    Matrix clone() {
        return Matrix(Rcpp::clone(x), Rcpp::clone(i), Rcpp::clone(p), Rcpp::clone(Dim));
    }

    // back to an R dgCMatrix sharing the same vectors (reference :387-394)
    Rcpp::S4 wrap() {
        Rcpp::S4 out(std::string("dgCMatrix"));
        out.slot("Dim") = Dim;
        out.slot("p") = p;
        out.slot("i") = i;
        out.slot("x") = x;
        return out;
    }

    // t(A) as a new Matrix (reference :375-385 calls back into R's Matrix::t; here a native
    // counting transpose with no R evaluation).  Like the reference's, this method allocates Rcpp
    // vectors and so belongs on the R main thread; only transpose_into() (plain std::vector
    // output, no R API) is safe off it.
    Matrix transpose() {
        std::vector<double> tx;
        std::vector<int> ti, tp;
        transpose_into(tx, ti, tp);
        Rcpp::IntegerVector d = Rcpp::IntegerVector::create(Dim[1], Dim[0]);
        return Matrix(Rcpp::NumericVector(tx.begin(), tx.end()), Rcpp::IntegerVector(ti.begin(), ti.end()),
                      Rcpp::IntegerVector(tp.begin(), tp.end()), d);
    }
};

}  // namespace RcppSparse

namespace Rcpp {
namespace traits {

// SEXP (S4 dgCMatrix) -> RcppSparse::Matrix for Rcpp::as<> and for
// input_parameter<RcppSparse::Matrix&> in the generated glue (reference :398-423).
// Slots are wrapped, not copied; the only check is slot presence.
template <>
class Exporter<RcppSparse::Matrix> {
    RcppSparse::Matrix held_;

public:
    Exporter(SEXP obj) : held_(Rcpp::S4(obj)) {}
    RcppSparse::Matrix get() { return held_; }
};

}  // namespace traits
}  // namespace Rcpp

#endif

// Compiles THIS repository's Rcpp layer (drop-in header + exported columnSums + glue) against the
// API-shaped mock in this directory and drives it the way R would: package init, routine lookup
// in the registration table, .Call with a dgCMatrix-like S4 object.
//   driver registered          -> prints the registered routine name and arity
//   driver missing_slot        -> S4 without `p`: the glue must turn the exception into an R error
//   driver kat                 -> Documentation.Rmd:213-216 matrix through .Call (needs a GPU)
#include "../../rcppsparse_amd/host/RcppSparse.h"

#include <cstdio>
#include <cstring>

extern "C" SEXP _RcppSparse_columnSums(SEXP);
extern "C" void R_init_RcppSparse(DllInfo*);

static SEXP dgc(bool with_p) {
    Rcpp::S4 A(std::string("dgCMatrix"));
    const double x[5] = {0.41, 0.35, 0.84, 0.37, 0.26};
    const int i[5] = {0, 2, 0, 1, 1}, p[6] = {0, 0, 1, 2, 4, 5}, dim[2] = {5, 5};
    A.slot("x") = Rcpp::NumericVector(x, x + 5);
    A.slot("i") = Rcpp::IntegerVector(i, i + 5);
    if (with_p) A.slot("p") = Rcpp::IntegerVector(p, p + 6);
    A.slot("Dim") = Rcpp::IntegerVector(dim, dim + 2);
    return A;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "registered";
    DllInfo dll = {0, true};
    R_init_RcppSparse(&dll);
    if (!dll.registered || dll.dynamic_symbols) return 10;
    if (mode == "registered") {
        std::printf("%s %d\n", dll.registered[0].name, dll.registered[0].numArgs);
        return (dll.registered[1].name == 0) ? 0 : 11;
    }
    typedef SEXP (*call1)(SEXP);
    call1 fn = (call1)dll.registered[0].fun;          // what .Call(`_RcppSparse_columnSums`, A) resolves to
    if (mode == "missing_slot") {
        SEXP r = fn(dgc(false));
        std::printf("%s\n", r->error.c_str());
        return r->error.empty() ? 12 : 0;
    }
    if (mode == "kat") {
        RcppSparse::Matrix M{Rcpp::S4(dgc(true))};
        if (M.cols() != 5 || M.n_nonzero() != 5) return 13;
        Rcpp::S4 back = M.wrap();                      // round trip shares the vectors
        if (!back.hasSlot("x") || RcppSparse::Matrix(back).x.storage() != M.x.storage()) return 14;
        RcppSparse::Matrix T = M.transpose();
        if (T.cols() != 5 || T.at(1, 3) != M.at(3, 1)) return 15;
        SEXP r = fn(dgc(true));
        if (!r->error.empty()) { std::printf("R error: %s\n", r->error.c_str()); return 16; }
        const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
        if (r->num->size() != 5 || std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 17;
        std::printf("columnSums via .Call ok\n");
        return 0;
    }
    return 1;
}

// host_seam.cpp -- Rcpp-free build of the host mirror, for tests.
//
// R and Rcpp are not in this image, so the Rcpp layer (host/RcppSparse.h,
// host/rpkg/src/columnSums.cpp, host/rpkg/src/rcpp_glue.cpp) cannot be compiled here.  Everything
// beneath that layer can: this file instantiates the same templates
// (rcppsparse_core::CscMatrix, column_sums_via_hip) over plain vector types
// that mimic the three Rcpp properties the class relies on -- copies share
// storage (by-reference semantics), construction from foreign memory is
// zero-copy, and `Vector(n)` is zero-filled -- and exports a small C surface
// that tests/test_host_mirror.py drives through ctypes.  columnSums() here is
// the same code the Rcpp build runs: Matrix& in, one call across the C ABI of
// include/rcppsparse_hip.h, numeric vector out.
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "columnsums_impl.hpp"
#include "rcppsparse_core.hpp"

namespace rcppsparse_seam {

// Stand-in for Rcpp::Vector<RTYPE>: shared storage, or a borrowed view.
template <class T>
class Vector {
public:
    Vector() : own_(std::make_shared<std::vector<T> >()), ptr_(0), n_(0) {}
    explicit Vector(std::size_t n) : own_(std::make_shared<std::vector<T> >(n, T())), ptr_(n ? &(*own_)[0] : 0), n_(n) {}
    Vector(T* borrowed, std::size_t n) : ptr_(borrowed), n_(n) {}   // zero-copy view (like wrapping a SEXP)
    T& operator[](std::size_t k) { return ptr_[k]; }
    const T& operator[](std::size_t k) const { return ptr_[k]; }
    T& operator()(std::size_t k) { return ptr_[k]; }
    std::size_t size() const { return n_; }
    T* begin() { return ptr_; }
    T* end() { return ptr_ + n_; }

private:
    std::shared_ptr<std::vector<T> > own_;
    T* ptr_;
    std::size_t n_;
};
typedef Vector<double> NumericVector;
typedef Vector<int> IntegerVector;

// Stand-in for Rcpp::NumericMatrix: column-major, zero-filled.
class NumericMatrix {
public:
    NumericMatrix() : nr_(0), nc_(0) {}
    NumericMatrix(std::size_t nr, std::size_t nc) : d_(nr * nc), nr_(nr), nc_(nc) {}
    double& operator()(std::size_t r, std::size_t c) { return d_[c * nr_ + r]; }
    std::size_t nrow() const { return nr_; }
    std::size_t ncol() const { return nc_; }
    NumericVector& data() { return d_; }

private:
    NumericVector d_;
    std::size_t nr_, nc_;
};

// Stand-in for Rcpp::S4: a bag of named slots.
struct S4 {
    std::map<std::string, NumericVector> num;
    std::map<std::string, IntegerVector> integer;
    bool hasSlot(const char* name) const { return num.count(name) || integer.count(name); }
};

struct Traits {
    typedef rcppsparse_seam::NumericVector NumVec;
    typedef rcppsparse_seam::IntegerVector IntVec;
    typedef rcppsparse_seam::NumericMatrix NumMat;
    static NumVec zeros(std::size_t n) { return NumVec(n); }
    static NumMat zeros(std::size_t r, std::size_t c) { return NumMat(r, c); }
    static NumVec num_slot(const S4& s, const char* name) { return s.num.find(name)->second; }
    static IntVec int_slot(const S4& s, const char* name) { return s.integer.find(name)->second; }
};

class Matrix : public rcppsparse_core::CscMatrix<Traits> {
public:
    typedef rcppsparse_core::CscMatrix<Traits> Base;
    Matrix() {}
    Matrix(NumericVector x, IntegerVector i, IntegerVector p, IntegerVector Dim) : Base(x, i, p, Dim) {}
    explicit Matrix(const S4& s) { assign_from_slots(s); }
    Matrix transpose() {
        std::vector<double> tx;
        std::vector<int> ti, tp;
        transpose_into(tx, ti, tp);
        NumericVector nx(tx.size());
        IntegerVector ni(ti.size()), np(tp.size()), nd(2);
        if (!tx.empty()) std::memcpy(&nx[0], &tx[0], tx.size() * sizeof(double));
        if (!ti.empty()) std::memcpy(&ni[0], &ti[0], ti.size() * sizeof(int));
        std::memcpy(&np[0], &tp[0], tp.size() * sizeof(int));
        nd[0] = Dim[1];
        nd[1] = Dim[0];
        return Matrix(nx, ni, np, nd);
    }
};

// The exported function, same signature shape as reference src/example.cpp:26.
NumericVector columnSums(Matrix& A) {
    return rcppsparse_core::column_sums_via_hip<Matrix, Traits>(A);
}

// The reference's own loop, verbatim in spirit, over THIS Matrix class: proves the
// InnerIterator of the mirror walks exactly the storage range the reference's does.
NumericVector columnSums_by_iterator(Matrix& A) {
    NumericVector sums(A.cols());
    for (std::size_t col = 0; col < A.cols(); ++col)
        for (Matrix::InnerIterator it(A, (int)col); it; ++it) sums(col) += it.value();
    return sums;
}

}  // namespace rcppsparse_seam

// ---------------------------------------------------------------------------
// C surface for ctypes
// ---------------------------------------------------------------------------
using namespace rcppsparse_seam;

namespace {
thread_local std::string g_seam_err;

Matrix view(const double* x, const int* i, const int* p, const int* dim, int nnz) {
    return Matrix(NumericVector(const_cast<double*>(x), (std::size_t)nnz),
                  IntegerVector(const_cast<int*>(i), i ? (std::size_t)nnz : 0),
                  IntegerVector(const_cast<int*>(p), (std::size_t)dim[1] + 1),
                  IntegerVector(const_cast<int*>(dim), 2));
}

template <class F>
int guarded(F f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {   // what END_RCPP does: exception -> error text
        g_seam_err = e.what();
        return 1;
    }
}
}  // namespace

extern "C" {

const char* seam_last_error(void) { return g_seam_err.c_str(); }

// columnSums(A) through the HIP shim (needs a GPU)
int seam_columnSums(const double* x, const int* i, const int* p, const int* dim, int nnz, double* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        NumericVector s = columnSums(A);
        for (std::size_t k = 0; k < s.size(); ++k) out[k] = s[k];
    });
}

// columnSums(A) with the R-level option twin spelled out: require_gpu = 1 / 0, or -1 = the environment decides
int seam_columnSums_opt(const double* x, const int* i, const int* p, const int* dim, int nnz, int require_gpu,
                        double* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        NumericVector s = rcppsparse_core::column_sums_via_hip<Matrix, Traits>(A, require_gpu);
        for (std::size_t k = 0; k < s.size(); ++k) out[k] = s[k];
    });
}

// ... and with the offload threshold's option twin: min_nnz >= 0 as options(RcppSparse.min_nnz = n), -1 = the
// environment (RCPPSPARSE_MIN_NNZ), else the measured default
int seam_columnSums_opt2(const double* x, const int* i, const int* p, const int* dim, int nnz, int require_gpu,
                         long long min_nnz, double* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        NumericVector s = rcppsparse_core::column_sums_via_hip<Matrix, Traits>(A, require_gpu, min_nnz);
        for (std::size_t k = 0; k < s.size(); ++k) out[k] = s[k];
    });
}

// 0 none, 1 hip, 2 cpu: the path the most recent columnSums took (last != 0) / a call would take now
int seam_backend(int last, int require_gpu) {
    return last ? rcppsparse_core::last_backend() : rcppsparse_core::choose_backend(require_gpu);
}
// ... for a matrix of `nnz` stored entries under the offload threshold (min_nnz as above)
int seam_backend_for(long long nnz, int require_gpu, long long min_nnz) {
    return rcppsparse_core::choose_backend(require_gpu, nnz, min_nnz);
}
long long seam_min_nnz(long long option) { return rcppsparse_core::min_nnz_setting(option); }

// the reference loop over the mirror's InnerIterator (CPU; tests iterator semantics)
int seam_columnSums_by_iterator(const double* x, const int* i, const int* p, const int* dim, int nnz,
                                double* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        NumericVector s = columnSums_by_iterator(A);
        for (std::size_t k = 0; k < s.size(); ++k) out[k] = s[k];
    });
}

// sizes as the class reports them: rows, cols, nrow, ncol, n_nonzero, InnerNNZs(col0)
int seam_sizes(const double* x, const int* i, const int* p, const int* dim, int nnz, int col0,
               unsigned int* out6) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        out6[0] = A.rows();
        out6[1] = A.cols();
        out6[2] = A.nrow();
        out6[3] = A.ncol();
        out6[4] = A.n_nonzero();
        out6[5] = dim[1] > 0 ? A.InnerNNZs(col0) : 0;
    });
}

// InnerIterator walk of one column: writes (row, value, col) triplets; returns count via *n
int seam_walk_column(const double* x, const int* i, const int* p, const int* dim, int nnz, int col,
                     int* rows, double* vals, int* cols, int* n) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        int k = 0;
        for (Matrix::InnerIterator it(A, col); it; ++it, ++k) {
            rows[k] = it.row();
            vals[k] = it.value();
            cols[k] = it.col();
        }
        *n = k;
    });
}

// mode 0: InnerIteratorInRange, mode 1: InnerIteratorNotInRange, mode 2: InnerRowIterator (col := row index)
int seam_walk_restricted(const double* x, const int* i, const int* p, const int* dim, int nnz, int col,
                         const unsigned int* s, int ns, int mode, int* idx, double* vals, int* n) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        std::vector<unsigned int> set(s, s + ns);
        int k = 0;
        if (mode == 0) {
            for (Matrix::InnerIteratorInRange it(A, col, set); it; ++it, ++k) { idx[k] = it.row(); vals[k] = it.value(); }
        } else if (mode == 1) {
            for (Matrix::InnerIteratorNotInRange it(A, col, set); it; ++it, ++k) { idx[k] = it.row(); vals[k] = it.value(); }
        } else {
            for (Matrix::InnerRowIterator it(A, col); it; ++it, ++k) { idx[k] = it.col(); vals[k] = it.value(); }
        }
        *n = k;
    });
}

// dense views and reductions of the header (CPU): which = 0 colSums, 1 rowSums, 2 colMeans, 3 rowMeans,
// 4 col(col0), 5 row(col0), 6 crossprod (ncol*ncol, column-major), 7 at(col0, col1)
int seam_dense(const double* x, const int* i, const int* p, const int* dim, int nnz, int which,
               int a, int b, double* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        NumericVector v;
        switch (which) {
            case 0: v = A.colSums(); break;
            case 1: v = A.rowSums(); break;
            case 2: v = A.colMeans(); break;
            case 3: v = A.rowMeans(); break;
            case 4: v = A.col(a); break;
            case 5: v = A.row(a); break;
            case 6: { NumericMatrix m = A.crossprod(); v = m.data(); break; }
            case 7: { out[0] = A.at(a, b); return; }
            default: throw std::invalid_argument("bad selector");
        }
        for (std::size_t k = 0; k < v.size(); ++k) out[k] = v[k];
    });
}

// transpose(): writes slots of t(A); tp has nrow+1 entries
int seam_transpose(const double* x, const int* i, const int* p, const int* dim, int nnz, double* tx,
                   int* ti, int* tp) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        Matrix T = A.transpose();
        for (int k = 0; k < nnz; ++k) { tx[k] = T.x[k]; ti[k] = T.i[k]; }
        for (int k = 0; k <= dim[0]; ++k) tp[k] = T.p[k];
    });
}

// sub-view clones and index helpers (RcppSparse.h:73-128, :198-215): `rows`/`cols` are index lists.
// which = 0: A(row0, cols) -> ncols values; 1: A(rows, col0) -> nrows values; 2: A(rows, cols) -> nrows x ncols
// (column-major); 3: A.col(cols) -> nrow x ncols; 4: A.row(rows) -> nrows x ncol; 5: A[index0];
// 6: InnerIndices(col0) -> count in out[0] then the indices; 7: emptyInnerIndices(col0) likewise.
int seam_subviews(const double* x, const int* i, const int* p, const int* dim, int nnz, int which,
                  const int* rows, int nrows, const int* cols, int ncols, int a0, double* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        IntegerVector R(const_cast<int*>(rows), (std::size_t)nrows), C(const_cast<int*>(cols), (std::size_t)ncols);
        if (which == 0) { NumericVector v = A(a0, C); for (std::size_t k = 0; k < v.size(); ++k) out[k] = v[k]; }
        else if (which == 1) { NumericVector v = A(R, a0); for (std::size_t k = 0; k < v.size(); ++k) out[k] = v[k]; }
        else if (which == 2) { NumericMatrix m = A(R, C); NumericVector& d = m.data(); for (std::size_t k = 0; k < d.size(); ++k) out[k] = d[k]; }
        else if (which == 3) { NumericMatrix m = A.col(C); NumericVector& d = m.data(); for (std::size_t k = 0; k < d.size(); ++k) out[k] = d[k]; }
        else if (which == 4) { NumericMatrix m = A.row(R); NumericVector& d = m.data(); for (std::size_t k = 0; k < d.size(); ++k) out[k] = d[k]; }
        else if (which == 5) { out[0] = A[a0]; }
        else if (which == 6 || which == 7) {
            std::vector<unsigned int> v = which == 6 ? A.InnerIndices(a0) : A.emptyInnerIndices(a0);
            out[0] = (double)v.size();
            for (std::size_t k = 0; k < v.size(); ++k) out[k + 1] = (double)v[k];
        } else throw std::invalid_argument("bad selector");
    });
}

int seam_is_appx_symmetric(const double* x, const int* i, const int* p, const int* dim, int nnz, int* out) {
    return guarded([&] {
        Matrix A = view(x, i, p, dim, nnz);
        *out = A.isAppxSymmetric() ? 1 : 0;
    });
}

// S4 construction: `mask` bit k set = slot k present (0 x, 1 i, 2 p, 3 Dim).  Returns 1 and
// sets the error text (the reference's message) when a slot is missing.
int seam_construct_from_s4(int mask) {
    return guarded([&] {
        S4 s;
        if (mask & 1) s.num["x"] = NumericVector(1);
        if (mask & 2) s.integer["i"] = IntegerVector(1);
        if (mask & 4) s.integer["p"] = IntegerVector(2);
        if (mask & 8) s.integer["Dim"] = IntegerVector(2);
        Matrix A(s);
        (void)A;
    });
}

// by-reference semantics: a Matrix built from vectors shares their storage
// (reference vignette Documentation.Rmd:335-347 mutates the R object through the class)
int seam_shares_storage(void) {
    NumericVector x(3);
    IntegerVector i(3), p(2), d(2);
    p[1] = 3; d[0] = 3; d[1] = 1;
    Matrix A(x, i, p, d);
    A.x[1] = 42.0;
    return x[1] == 42.0 ? 1 : 0;
}

}  // extern "C"

// rcppsparse_core.hpp -- storage-agnostic core of the RcppSparse::Matrix drop-in.
//
// The reference class (inst/include/RcppSparse.h:25-396) is written directly
// against Rcpp vectors, so nothing in it can be compiled or unit-tested without
// R.  Here the same public surface is a template over a small `Traits` bundle
// (numeric vector, integer vector, dense matrix, how to allocate them), which
// gives two instantiations from one source:
//   * RcppSparse::Matrix            -- Traits = Rcpp types   (host/RcppSparse.h, needs R)
//   * rcppsparse_seam::Matrix       -- Traits = plain views  (host/host_seam.cpp, g++ only)
// Pure C++14, no HIP, no R: downstream `sourceCpp` users that only include the
// header keep compiling on machines without ROCm.
//
// Member names, argument meaning and return types follow the reference so that
// user code written against it compiles unchanged; each member cites the
// reference lines it stands in for.  Implementations are new (e.g. element
// lookup is a binary search over the sorted row indices, the restricted column
// iterators are a two-pointer merge written to the *documented* intent, and
// transpose() is a native counting transpose instead of a callback into R).
#ifndef RCPPSPARSE_CORE_HPP
#define RCPPSPARSE_CORE_HPP

#include <algorithm>
#include <cstddef>
#include <stdexcept>
#include <vector>

namespace rcppsparse_core {

// Message the reference throws when an S4 object lacks a dgCMatrix slot
// (RcppSparse.h:35-36 and :409-410).  Kept byte-identical.
inline const char* missing_slot_message() {
    return "Cannot construct RcppSparse::Matrix from this S4 object";
}

// True when `s` (anything with hasSlot(name)) carries the four dgCMatrix slots.
template <class S4Like>
inline bool has_dgc_slots(const S4Like& s) {
    return s.hasSlot("x") && s.hasSlot("p") && s.hasSlot("i") && s.hasSlot("Dim");
}

template <class Traits>
class CscMatrix {
public:
    typedef typename Traits::NumVec NumVec;
    typedef typename Traits::IntVec IntVec;
    typedef typename Traits::NumMat NumMat;

    // the four dgCMatrix slots, public like the reference (RcppSparse.h:29-30)
    NumVec x;
    IntVec i, p, Dim;

    // ---- construction (RcppSparse.h:33-42) ---------------------------------
    CscMatrix() {}
    CscMatrix(NumVec x_, IntVec i_, IntVec p_, IntVec Dim_) : x(x_), i(i_), p(p_), Dim(Dim_) {}
    // from an S4-like object: slot presence is the only validation (as in :34-41).
    // The concrete classes expose this as their `Matrix(const S4&)` constructor.
    template <class S4Like>
    void assign_from_slots(const S4Like& s) {
        if (!has_dgc_slots(s)) throw std::invalid_argument(missing_slot_message());
        x = Traits::num_slot(s, "x");
        i = Traits::int_slot(s, "i");
        p = Traits::int_slot(s, "p");
        Dim = Traits::int_slot(s, "Dim");
    }

    // ---- sizes and raw slot access (RcppSparse.h:44-51) --------------------
    unsigned int rows() { return Dim[0]; }
    unsigned int cols() { return Dim[1]; }
    unsigned int nrow() { return Dim[0]; }
    unsigned int ncol() { return Dim[1]; }
    unsigned int n_nonzero() { return x.size(); }
    NumVec& nonzeros() { return x; }
    IntVec& innerIndexPtr() { return i; }
    IntVec& outerIndexPtr() { return p; }

    // number of stored entries in a column (RcppSparse.h:357-359)
    unsigned int InnerNNZs(int col) { return p[col + 1] - p[col]; }

    // ---- const column cursor: THE hot-path iterator (RcppSparse.h:218-233) --
    // Walks the half-open storage range [p[col], p[col+1]); all state is int.
    class InnerIterator {
    public:
        InnerIterator(CscMatrix& m, int col) : m_(m), col_(col), pos_(m.p[col]), end_(m.p[col + 1]) {}
        operator bool() const { return pos_ < end_; }
        InnerIterator& operator++() {
            ++pos_;
            return *this;
        }
        const double& value() const { return m_.x[pos_]; }
        int row() const { return m_.i[pos_]; }
        int col() const { return col_; }

    private:
        CscMatrix& m_;
        int col_, pos_, end_;
    };

    // ---- element and sub-view access (RcppSparse.h:63-128) -----------------
    // rows inside a column are ascending in a valid dgCMatrix: binary search
    double at(int row, int col) const {
        int lo = p[col], hi = p[col + 1];
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (i[mid] < row) lo = mid + 1; else hi = mid;
        }
        return (lo < p[col + 1] && i[lo] == row) ? x[lo] : 0.0;
    }
    double operator()(int row, int col) const { return at(row, col); }
    double operator[](int index) const { return x[index]; }

    NumVec operator()(int row, IntVec& cols_) {
        NumVec out = Traits::zeros(cols_.size());
        for (int k = 0; k < (int)cols_.size(); ++k) out[k] = at(row, cols_[k]);
        return out;
    }
    NumVec operator()(IntVec& rows_, int col) {
        NumVec out = Traits::zeros(rows_.size());
        for (int k = 0; k < (int)rows_.size(); ++k) out[k] = at(rows_[k], col);
        return out;
    }
    NumMat operator()(IntVec& rows_, IntVec& cols_) {
        NumMat out = Traits::zeros(rows_.size(), cols_.size());
        for (int c = 0; c < (int)cols_.size(); ++c)
            for (int r = 0; r < (int)rows_.size(); ++r) out(r, c) = at(rows_[r], cols_[c]);
        return out;
    }

    // dense copy of one column / several columns
    NumVec col(int c) {
        NumVec out = Traits::zeros(Dim[0]);
        for (InnerIterator it(*this, c); it; ++it) out[it.row()] = it.value();
        return out;
    }
    NumMat col(IntVec& cs) {
        NumMat out = Traits::zeros(Dim[0], cs.size());
        for (int k = 0; k < (int)cs.size(); ++k)
            for (InnerIterator it(*this, cs[k]); it; ++it) out(it.row(), k) = it.value();
        return out;
    }
    // dense copy of one row / several rows
    NumVec row(int r) {
        NumVec out = Traits::zeros(Dim[1]);
        for (int c = 0; c < Dim[1]; ++c) out[c] = at(r, c);
        return out;
    }
    NumMat row(IntVec& rs) {
        NumMat out = Traits::zeros(rs.size(), Dim[1]);
        for (int c = 0; c < Dim[1]; ++c)
            for (int k = 0; k < (int)rs.size(); ++k) out(k, c) = at(rs[k], c);
        return out;
    }

    // ---- reductions (RcppSparse.h:131-156); CPU, header-inline ---------------
    // Same arithmetic as the reference: one accumulator per output, plain +=
    // in storage order.  (The GPU path lives behind the *exported* columnSums,
    // see host/columnsums_impl.hpp; a header-only consumer has no HIP.)
    NumVec colSums() {
        NumVec out = Traits::zeros(Dim[1]);
        for (int c = 0; c < Dim[1]; ++c)
            for (InnerIterator it(*this, c); it; ++it) out[c] += it.value();
        return out;
    }
    NumVec rowSums() {
        NumVec out = Traits::zeros(Dim[0]);
        for (int c = 0; c < Dim[1]; ++c)
            for (InnerIterator it(*this, c); it; ++it) out[it.row()] += it.value();
        return out;
    }
    NumVec colMeans() {
        NumVec out = colSums();
        for (int k = 0; k < (int)out.size(); ++k) out[k] = out[k] / Dim[0];
        return out;
    }
    NumVec rowMeans() {
        NumVec out = rowSums();
        for (int k = 0; k < (int)out.size(); ++k) out[k] = out[k] / Dim[1];
        return out;
    }

    // t(A) %*% A as a dense ncol x ncol matrix (RcppSparse.h:159-194): sparse
    // dot product of every column pair by a sorted merge; columns in parallel
    // under OpenMP when the consumer enables it.
    NumMat crossprod() {
        const int n = Dim[1];
        NumMat out = Traits::zeros(n, n);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8)
#endif
        for (int a = 0; a < n; ++a) {
            for (int b = a; b < n; ++b) {
                int ja = p[a], jb = p[b];
                const int ea = p[a + 1], eb = p[b + 1];
                double dot = 0.0;
                while (ja < ea && jb < eb) {
                    const int ra = i[ja], rb = i[jb];
                    if (ra == rb) {
                        dot += x[ja] * x[jb];
                        ++ja;
                        ++jb;
                    } else if (ra < rb) {
                        ++ja;
                    } else {
                        ++jb;
                    }
                }
                out(a, b) = dot;
                out(b, a) = dot;
            }
        }
        return out;
    }

    // ---- index helpers (RcppSparse.h:198-215) ------------------------------
    std::vector<unsigned int> InnerIndices(int col) {
        std::vector<unsigned int> v;
        v.reserve(p[col + 1] - p[col]);
        for (InnerIterator it(*this, col); it; ++it) v.push_back((unsigned int)it.row());
        return v;
    }
    // rows of the column that hold no stored entry
    std::vector<unsigned int> emptyInnerIndices(int col) {
        std::vector<unsigned int> v;
        unsigned int next = 0;
        for (InnerIterator it(*this, col); it; ++it) {
            for (; next < (unsigned int)it.row(); ++next) v.push_back(next);
            next = (unsigned int)it.row() + 1;
        }
        for (; next < (unsigned int)Dim[0]; ++next) v.push_back(next);
        return v;
    }

    // ---- restricted column cursors (RcppSparse.h:238-321) ------------------
    // Stored entries of column `col` whose row IS in the ascending set `s`.
    // Written to the documented intent; the reference's out-of-bounds probes
    // (:242, :299) and off-by-one (:282) are not reproduced.
    class InnerIteratorInRange {
    public:
        InnerIteratorInRange(CscMatrix& m, int col, std::vector<unsigned int>& s)
            : m_(m), s_(s), col_(col), pos_(m.p[col]), end_(m.p[col + 1]), k_(0) {
            settle();
        }
        operator bool() const { return pos_ < end_ && k_ < s_.size(); }
        InnerIteratorInRange& operator++() {
            ++pos_;
            ++k_;
            settle();
            return *this;
        }
        const double& value() const { return m_.x[pos_]; }
        int row() const { return m_.i[pos_]; }
        int col() const { return col_; }

    private:
        void settle() {   // advance both cursors to the next common row
            while (pos_ < end_ && k_ < s_.size()) {
                const unsigned int r = (unsigned int)m_.i[pos_];
                if (r == s_[k_]) return;
                if (r < s_[k_]) ++pos_; else ++k_;
            }
        }
        CscMatrix& m_;
        const std::vector<unsigned int>& s_;
        int col_, pos_, end_;
        std::size_t k_;
    };

    // Stored entries of column `col` whose row is NOT in the ascending set `s`.
    class InnerIteratorNotInRange {
    public:
        InnerIteratorNotInRange(CscMatrix& m, int col, std::vector<unsigned int>& s)
            : m_(m), s_(s), col_(col), pos_(m.p[col]), end_(m.p[col + 1]), k_(0) {
            settle();
        }
        operator bool() const { return pos_ < end_; }
        InnerIteratorNotInRange& operator++() {
            ++pos_;
            settle();
            return *this;
        }
        const double& value() const { return m_.x[pos_]; }
        int row() const { return m_.i[pos_]; }
        int col() const { return col_; }

    private:
        void settle() {   // skip stored entries whose row appears in s
            while (pos_ < end_) {
                const unsigned int r = (unsigned int)m_.i[pos_];
                while (k_ < s_.size() && s_[k_] < r) ++k_;
                if (k_ < s_.size() && s_[k_] == r) ++pos_; else return;
            }
        }
        CscMatrix& m_;
        std::vector<unsigned int> s_;   // own copy, like the reference (:317)
        int col_, pos_, end_;
        std::size_t k_;
    };

    // ---- row cursor (RcppSparse.h:324-354) ---------------------------------
    // Stored entries of row `j` in ascending column order.  Like the
    // reference's it is O(nnz) to build (it scans i[] once); documented there
    // as inefficient.  Unlike the reference it honours j != 0.
    class InnerRowIterator {
    public:
        InnerRowIterator(CscMatrix& m, int j) : m_(m), row_(j), k_(0) {
            const int n = m.Dim[1];
            for (int c = 0; c < n; ++c) {
                int lo = m.p[c], hi = m.p[c + 1];
                while (lo < hi) {
                    const int mid = lo + (hi - lo) / 2;
                    if (m.i[mid] < j) lo = mid + 1; else hi = mid;
                }
                if (lo < m.p[c + 1] && m.i[lo] == j) {
                    pos_.push_back(lo);
                    col_.push_back(c);
                }
            }
        }
        operator bool() const { return k_ < pos_.size(); }
        InnerRowIterator& operator++() {
            ++k_;
            return *this;
        }
        int col() { return col_[k_]; }
        int row() { return row_; }
        double& value() const { return m_.x[pos_[k_]]; }

    private:
        CscMatrix& m_;
        int row_;
        std::size_t k_;
        std::vector<int> pos_, col_;
    };

    // square, and row 0 mirrors column 0 entry for entry (RcppSparse.h:362-373:
    // the reference's cheap necessary condition, not a full symmetry test)
    bool isAppxSymmetric() {
        if (Dim[0] != Dim[1]) return false;
        InnerIterator c(*this, 0);
        InnerRowIterator r(*this, 0);
        for (; c && r; ++c, ++r)
            if (c.row() != r.col() || c.value() != r.value()) return false;
        return !c && !r;
    }

    // ---- transpose as raw slots (used by the Traits-specific transpose()) --
    // Counting transpose: O(nnz + nrow + ncol), rows come out ascending.
    void transpose_into(std::vector<double>& tx, std::vector<int>& ti, std::vector<int>& tp) {
        const int nr = Dim[0], nc = Dim[1];
        const int nnz = (int)x.size();
        tp.assign((std::size_t)nr + 1, 0);
        for (int k = 0; k < nnz; ++k) ++tp[(std::size_t)i[k] + 1];
        for (int r = 0; r < nr; ++r) tp[(std::size_t)r + 1] += tp[r];
        std::vector<int> fill(tp.begin(), tp.end() - 1);
        tx.resize(nnz);
        ti.resize(nnz);
        for (int c = 0; c < nc; ++c)
            for (InnerIterator it(*this, c); it; ++it) {
                const int dst = fill[it.row()]++;
                tx[dst] = it.value();
                ti[dst] = c;
            }
    }
};

}  // namespace rcppsparse_core
#endif

/*
 * oracle/colsums_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the RcppSparse column-iteration hot path, over raw
 * pointers instead of Rcpp vectors.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path
 * (rcppsparse_amd/csrc) never links or calls it.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or fixtures for
 * columnSums, and it cannot be compiled here (needs
 * Rcpp/R headers that the image lacks), so there is no oracle/_ref build.  The
 * only reference-derived known answer is the literal 5x5 matrix printed in
 * vignettes/Documentation.Rmd:213-216; everything else is pinned by this
 * restatement cross-checked against SciPy.  => "parity unpinned" by the
 * reference's own tests; pinned by KAT + SciPy cross-check only.
 *
 * Build: see oracle/Makefile.  Must be compiled WITHOUT -ffast-math /
 * -fassociative-math so the adds stay sequential, one rounding per add.
 *
 * What is restated (reference file:line):
 *   src/example.cpp:26-32                 columnSums()
 *   inst/include/RcppSparse.h:218-233     Matrix::InnerIterator
 *   inst/include/RcppSparse.h:44-48       rows()/cols()/n_nonzero()
 *   inst/include/RcppSparse.h:131-137     Matrix::colSums()   (same sums, no iterator)
 *   inst/include/RcppSparse.h:145-150     Matrix::colMeans()
 *   inst/include/RcppSparse.h:138-144     Matrix::rowSums()   (scatter form; "next" row f1)
 *   inst/include/RcppSparse.h:151-156     Matrix::rowMeans()
 */
#include <stddef.h>
#include <stdint.h>

/* A dgCMatrix seen as raw slots (RcppSparse.h:29-30: x, i, p, Dim). */
typedef struct {
    const double  *x;
    const int32_t *i;
    const int32_t *p;
    int32_t        dim[2];
} oracle_csc;

/* RcppSparse.h:218-233 -- const column cursor; all state is 32-bit int (:232). */
typedef struct {
    const oracle_csc *m;
    int32_t col_, index, max_index;
} oracle_inner_it;

static void it_begin(oracle_inner_it *it, const oracle_csc *m, int32_t col) {
    /* :220  index(ptr.p[col]), max_index(ptr.p[col + 1]) */
    it->m = m;
    it->col_ = col;
    it->index = m->p[col];
    it->max_index = m->p[col + 1];
}
static int it_valid(const oracle_inner_it *it) { return it->index < it->max_index; } /* :221 */
static void it_next(oracle_inner_it *it) { ++it->index; }                             /* :222-225 */
static double it_value(const oracle_inner_it *it) { return it->m->x[it->index]; }     /* :226 */

/*
 * src/example.cpp:26-32.  `sums` plays the zero-filled NumericVector(A.cols())
 * of :27; the double loop is :28-30; accumulation is a plain `+=` into
 * sums[col] in ascending storage order starting from +0.0.
 */
void oracle_column_sums(const double *x, const int32_t *i, const int32_t *p,
                        int32_t nrow, int32_t ncol, double *sums) {
    oracle_csc A = {x, i, p, {nrow, ncol}};
    uint32_t cols = (uint32_t)A.dim[1]; /* RcppSparse.h:45 cols() -> unsigned int */
    for (size_t col = 0; col < cols; ++col) sums[col] = 0.0; /* example.cpp:27 */
    for (size_t col = 0; col < cols; ++col) {                /* example.cpp:28 */
        oracle_inner_it it;
        for (it_begin(&it, &A, (int32_t)col); it_valid(&it); it_next(&it)) /* :29 */
            sums[col] += it_value(&it);                                     /* :30 */
    }
}

/* RcppSparse.h:131-137 -- same reduction, direct p/x loop with int indices. */
void oracle_col_sums(const double *x, const int32_t *p, int32_t ncol, double *sums) {
    for (int32_t col = 0; col < ncol; ++col) sums[col] = 0.0;
    for (int32_t col = 0; col < ncol; ++col)
        for (int32_t j = p[col]; j < p[col + 1]; ++j)
            sums[col] += x[j];
}

/* RcppSparse.h:145-150 -- colSums() then divide each by Dim[0]. */
void oracle_col_means(const double *x, const int32_t *p, int32_t nrow, int32_t ncol,
                      double *means) {
    oracle_col_sums(x, p, ncol, means);
    for (int32_t c = 0; c < ncol; ++c) means[c] = means[c] / nrow;
}

/* RcppSparse.h:138-144 -- scatter-add by row index, column-major visiting order. */
void oracle_row_sums(const double *x, const int32_t *i, const int32_t *p,
                     int32_t nrow, int32_t ncol, double *sums) {
    for (int32_t r = 0; r < nrow; ++r) sums[r] = 0.0;
    for (int32_t col = 0; col < ncol; ++col)
        for (int32_t j = p[col]; j < p[col + 1]; ++j)
            sums[i[j]] += x[j];
}

/* The same scatter loop (RcppSparse.h:141-143) continued over a further run of n stored entries, in storage order:
 * calling it on consecutive slabs of x / i is the whole-matrix loop term for term.  abs_sums (may be NULL)
 * collects Sum|x| per row, the scale of the parity tolerance. */
void oracle_row_sums_accumulate(const double *x, const int32_t *i, int64_t n, double *sums, double *abs_sums) {
    for (int64_t j = 0; j < n; ++j) {
        sums[i[j]] += x[j];
        if (abs_sums) abs_sums[i[j]] += x[j] < 0 ? -x[j] : x[j];
    }
}

/* RcppSparse.h:151-156 -- rowSums() then divide each by Dim[1]. */
void oracle_row_means(const double *x, const int32_t *i, const int32_t *p,
                      int32_t nrow, int32_t ncol, double *means) {
    oracle_row_sums(x, i, p, nrow, ncol, means);
    for (int32_t r = 0; r < nrow; ++r) means[r] = means[r] / ncol;
}

/*
 * Per-column 1-norms, Sum_j |x_j|: the scale of the parity tolerance
 * (|gpu - ref| <= 1e-12 * Sum|x|).  Not in the reference.
 */
void oracle_column_abs_sums(const double *x, const int32_t *p, int32_t ncol, double *sums) {
    for (int32_t col = 0; col < ncol; ++col) {
        double s = 0.0;
        for (int32_t j = p[col]; j < p[col + 1]; ++j) s += (x[j] < 0 ? -x[j] : x[j]);
        sums[col] = s;
    }
}

/*
 * Generic column reduction ("next" row f3): the InnerIterator loop of example.cpp:28-30
 * with a different body, as the vignette's column loops do (Documentation.Rmd:303-312).
 * op 0: acc += v;  op 1: acc += v * v;  op 2: acc += |v|;  op 3: if (v > acc) acc = v from -Inf;
 * op 4: if (v < acc) acc = v from +Inf;  op 5: number of stored entries (InnerNNZs, :357-359).
 */
void oracle_column_reduce(const double *x, const int32_t *p, int32_t ncol, int op, double *out) {
    oracle_csc A = {x, NULL, p, {0, ncol}};
    for (int32_t col = 0; col < ncol; ++col) {
        double acc = (op == 3) ? -__builtin_huge_val() : (op == 4) ? __builtin_huge_val() : 0.0;
        oracle_inner_it it;
        for (it_begin(&it, &A, col); it_valid(&it); it_next(&it)) {
            const double v = it_value(&it);
            if (op == 3) { if (v > acc) acc = v; }
            else if (op == 4) { if (v < acc) acc = v; }
            else if (op == 5) acc += 1.0;
            else acc += (op == 1) ? v * v : (op == 2) ? (v < 0 ? -v : v) : v;
        }
        out[col] = acc;
    }
}

/*
 * Row-restricted column sums ("next" row f4): the column loop driven by
 * Matrix::InnerIteratorInRange / InnerIteratorNotInRange (RcppSparse.h:238-321), restated to
 * their documented intent -- visit the stored entries of the column whose row is / is not in
 * the ascending set s (here a bitmap), in storage order.  The reference's constructors probe
 * i[] and s[] out of bounds on empty inputs (:242, :299); that is not reproduced.
 */
void oracle_column_sums_in_rows(const double *x, const int32_t *i, const int32_t *p, int32_t ncol,
                                const uint32_t *bitmap, int complement, double *out) {
    for (int32_t col = 0; col < ncol; ++col) {
        double acc = 0.0;
        for (int32_t j = p[col]; j < p[col + 1]; ++j) {
            const int in = (int)((bitmap[i[j] >> 5] >> (i[j] & 31)) & 1u);
            if (in != (complement != 0)) acc += x[j];
        }
        out[col] = acc;
    }
}

/*
 * Matrix::crossprod (RcppSparse.h:159-194): dense ncol x ncol t(A) %*% A, column-major.  For
 * every pair col1 <= col2 the two (ascending) row lists are merged and x1 * x2 is added for
 * the common rows in ascending order; the diagonal adds x*x; res(col2, col1) mirrors
 * res(col1, col2).  The reference's inner do/while peeks at i[] one past a column's end before
 * testing the bound (:180-186); the bound is tested first here, the visited pairs are the same.
 */
void oracle_crossprod(const double *x, const int32_t *i, const int32_t *p, int32_t ncol, double *res) {
    for (int64_t k = 0; k < (int64_t)ncol * ncol; ++k) res[k] = 0.0;
    for (int32_t col1 = 0; col1 < ncol; ++col1) {
        for (int32_t col2 = col1; col2 < ncol; ++col2) {
            double acc = 0.0;
            if (col1 == col2) {
                for (int32_t j = p[col1]; j < p[col1 + 1]; ++j) acc += x[j] * x[j];
            } else {
                int32_t a = p[col1], amax = p[col1 + 1], b = p[col2], bmax = p[col2 + 1];
                while (a < amax && b < bmax) {
                    const int32_t r1 = i[a], r2 = i[b];
                    if (r1 == r2) { acc += x[a] * x[b]; ++a; ++b; }
                    else if (r1 < r2) ++a;
                    else ++b;
                }
            }
            res[(int64_t)col2 * ncol + col1] = acc;
            res[(int64_t)col1 * ncol + col2] = acc;
        }
    }
}

/*
 * Synthetic value generators shared bit-for-bit with the device generator in
 * rcppsparse_amd/csrc (integer hash -> exactly representable double), so the
 * host can rebuild any slice of a device-generated x[] without copying it.
 * splitmix64 finaliser keyed by (seed, global nnz index).
 *   kind 0: "rsparsematrix-like" signed values with two decimals in [-5.10, 5.10]
 *           (sum of four 8-bit uniforms, centred, /100 -- a bell-shaped stand-in
 *           for Matrix::rsparsematrix's rounded rnorm default)
 *   kind 1: all-positive U(0,1) with 53 random bits
 */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

double oracle_gen_value(uint64_t seed, uint64_t idx, int kind) {
    uint64_t h = mix64(seed * 0xD1342543DE82EF95ull + idx);
    if (kind == 1) return (double)(h >> 11) * 0x1.0p-53;
    int s = (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) + (int)((h >> 24) & 255);
    return (double)(s - 510) / 100.0;
}

void oracle_gen_values(double *x, uint64_t n, uint64_t seed, uint64_t first_idx, int kind) {
    for (uint64_t k = 0; k < n; ++k) x[k] = oracle_gen_value(seed, first_idx + k, kind);
}

/* Twin of the device generator gen_row_indices_kernel (integer-only stratified rows). */
void oracle_gen_row_indices(int32_t *i, const int32_t *p, int32_t nrow, int32_t c_first,
                            int32_t c_last, uint64_t seed) {
    /* writes i[p[c] - p[c_first] + r] for c in [c_first, c_last) */
    const int32_t base = p[c_first];
    for (int32_t c = c_first; c < c_last; ++c) {
        const int32_t lo = p[c], k = p[c + 1] - p[c];
        for (int32_t r = 0; r < k; ++r) {
            const uint64_t h = mix64(seed * 0xD1342543DE82EF95ull + 0x5bd1e995ull + (uint64_t)(lo + r));
            const uint64_t s0 = (uint64_t)r * (uint64_t)nrow / (uint64_t)k;
            const uint64_t s1 = (uint64_t)(r + 1) * (uint64_t)nrow / (uint64_t)k;
            const uint64_t width = s1 > s0 ? s1 - s0 : 1;
            uint64_t row = s0 + (((h >> 32) * width) >> 32);
            if (row >= (uint64_t)nrow) row = (uint64_t)nrow - 1;
            i[lo - base + r] = (int32_t)row;
        }
    }
}

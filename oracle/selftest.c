/* oracle/selftest.c -- TEST INFRASTRUCTURE.  Runs the oracle's loops over the vignette KAT
 * and a few thousand random ragged matrices (empty columns, empty matrix, single column) in a
 * build with AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on
 * this pool, so the CPU side is where they run):
 *     gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer selftest.c colsums_oracle.c -o selftest
 * Exit code 0 = clean. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void oracle_column_sums(const double *, const int32_t *, const int32_t *, int32_t, int32_t, double *);
void oracle_col_sums(const double *, const int32_t *, int32_t, double *);
void oracle_col_means(const double *, const int32_t *, int32_t, int32_t, double *);
void oracle_row_sums(const double *, const int32_t *, const int32_t *, int32_t, int32_t, double *);
void oracle_row_means(const double *, const int32_t *, const int32_t *, int32_t, int32_t, double *);
void oracle_column_abs_sums(const double *, const int32_t *, int32_t, double *);
void oracle_column_reduce(const double *, const int32_t *, int32_t, int, double *);
void oracle_column_sums_in_rows(const double *, const int32_t *, const int32_t *, int32_t, const uint32_t *, int, double *);
void oracle_crossprod(const double *, const int32_t *, const int32_t *, int32_t, double *);
void oracle_gen_values(double *, uint64_t, uint64_t, uint64_t, int);
void oracle_gen_row_indices(int32_t *, const int32_t *, int32_t, int32_t, int32_t, uint64_t);

static uint64_t rng = 88172645463325252ull;
static uint32_t rnd(uint32_t n) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng % n); }

int main(void) {
    const double x5[5] = {0.41, 0.35, 0.84, 0.37, 0.26};
    const int32_t i5[5] = {0, 2, 0, 1, 1}, p5[6] = {0, 0, 1, 2, 4, 5};
    const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
    double got[5];
    oracle_column_sums(x5, i5, p5, 5, 5, got);
    if (memcmp(got, want, sizeof want)) { fprintf(stderr, "KAT mismatch\n"); return 1; }

    for (int iter = 0; iter < 3000; ++iter) {
        const int32_t ncol = (int32_t)rnd(40), nrow = 1 + (int32_t)rnd(60);
        int32_t *p = malloc(sizeof(int32_t) * ((size_t)ncol + 1));
        p[0] = 0;
        for (int32_t c = 0; c < ncol; ++c) {
            int32_t k = (int32_t)rnd(4) == 0 ? 0 : (int32_t)rnd((uint32_t)nrow + 1);
            p[c + 1] = p[c] + k;
        }
        const int32_t nnz = p[ncol];
        double *x = malloc(sizeof(double) * ((size_t)nnz + 1));
        int32_t *ri = malloc(sizeof(int32_t) * ((size_t)nnz + 1));
        oracle_gen_values(x, (uint64_t)nnz, (uint64_t)iter, 7, iter & 1);
        if (ncol) oracle_gen_row_indices(ri, p, nrow, 0, ncol, (uint64_t)iter);
        for (int32_t c = 0; c < ncol; ++c)          /* rows ascending and distinct per column */
            for (int32_t j = p[c] + 1; j < p[c + 1]; ++j)
                if (ri[j] <= ri[j - 1] || ri[j] >= nrow) { fprintf(stderr, "bad rows\n"); return 1; }
        double *a = malloc(sizeof(double) * ((size_t)ncol + 1)), *b = malloc(sizeof(double) * ((size_t)ncol + 1));
        double *r = malloc(sizeof(double) * (size_t)nrow), *cp = malloc(sizeof(double) * ((size_t)ncol * ncol + 1));
        uint32_t *bits = calloc(((size_t)nrow + 31) / 32, sizeof(uint32_t));
        for (int32_t q = 0; q < nrow; q += 3) bits[q >> 5] |= 1u << (q & 31);
        oracle_column_sums(x, ri, p, nrow, ncol, a);
        oracle_col_sums(x, p, ncol, b);
        if (memcmp(a, b, sizeof(double) * (size_t)ncol)) { fprintf(stderr, "iterator vs direct loop\n"); return 1; }
        oracle_col_means(x, p, nrow, ncol, b);
        oracle_row_sums(x, ri, p, nrow, ncol, r);
        oracle_row_means(x, ri, p, nrow, ncol, r);
        oracle_column_abs_sums(x, p, ncol, b);
        for (int op = 0; op <= 5; ++op) oracle_column_reduce(x, p, ncol, op, b);
        oracle_column_sums_in_rows(x, ri, p, ncol, bits, 0, a);
        oracle_column_sums_in_rows(x, ri, p, ncol, bits, 1, b);
        oracle_crossprod(x, ri, p, ncol, cp);
        free(p); free(x); free(ri); free(a); free(b); free(r); free(cp); free(bits);
    }
    printf("oracle selftest ok\n");
    return 0;
}

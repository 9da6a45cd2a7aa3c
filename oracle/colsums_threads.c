/*
 * colsums_threads.c -- the reference loop of src/example.cpp:26-32 spread over host threads.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (same rules as colsums_oracle.c: never part of the product
 * path).  The reference's columnSums path is single-threaded (its only OpenMP is in crossprod,
 * inst/include/RcppSparse.h:161-163); this file exists so that bench.py can report, next to
 * the faithful 1-thread baseline, what an OpenMP `parallel for` over the columns reaches on the
 * GPU box's host cores ("optionally also an all-cores OpenMP run").  Every
 * column is still summed sequentially in storage order from +0.0, so the results are
 * bit-identical to oracle_column_sums.
 */
#include <stddef.h>
#include <stdint.h>

#include <omp.h>

void oracle_threads_column_sums(const double *x, const int32_t *p, int32_t ncol, double *sums,
                                int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 512)
    for (int32_t col = 0; col < ncol; ++col) {
        double acc = 0.0;                                    /* example.cpp:27 */
        for (int32_t j = p[col]; j < p[col + 1]; ++j)        /* RcppSparse.h:220-221 */
            acc += x[j];                                     /* example.cpp:30 */
        sums[col] = acc;
    }
}

/* defined in colsums_oracle.c */
double oracle_gen_value(uint64_t seed, uint64_t idx, int kind);

/* parallel fill (first touch spreads the pages over the NUMA nodes of the threads) */
void oracle_threads_gen_values(double *x, uint64_t n, uint64_t seed, uint64_t first_idx, int kind,
                               int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (uint64_t k = 0; k < n; ++k) x[k] = oracle_gen_value(seed, first_idx + k, kind);
}

int oracle_threads_max(void) { return omp_get_max_threads(); }

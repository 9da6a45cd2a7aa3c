/* Plain-C caller of the C ABI (no Python, no torch, no C++): what a foreign host
 * language sees.  Builds the 5x5 matrix of reference vignettes/Documentation.Rmd:213-216,
 * calls the one-shot and the handle entry points, checks the exact expected bits, then a
 * 3e6-nnz ragged matrix against a plain C loop, then (round 6) the single-process multi-GPU
 * handle in every launch x gather (the reference loop restated inline here so
 * this file has no dependency on oracle/).  Exit code 0 = pass.
 *   gcc cabi_smoke.c -I../../include -L../../rcppsparse_amd -lrcppsparse_hip -Wl,-rpath,... */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcppsparse_hip.h"

static int fail(const char *what) {
    fprintf(stderr, "FAIL %s: %s\n", what, rsp_last_error());
    return 1;
}

int main(void) {
    int ndev = -1;
    if (rsp_device_count(&ndev) != RSP_OK) return fail("device_count");
    printf("%s, %d device(s)\n", rsp_version(), ndev);
    if (ndev <= 0) { fprintf(stderr, "no GPU: the library has no CPU fallback\n"); return 2; }

    const double x[5] = {0.41, 0.35, 0.84, 0.37, 0.26};
    const int32_t i[5] = {0, 2, 0, 1, 1};
    const int32_t p[6] = {0, 0, 1, 2, 4, 5};
    const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
    double got[5];
    if (rsp_column_sums_host(x, p, 5, 5, got, 0) != RSP_OK) return fail("column_sums_host");
    if (memcmp(got, want, sizeof want) != 0) { fprintf(stderr, "KAT bits differ\n"); return 1; }

    rsp_csc_t h = NULL;
    if (rsp_csc_upload(x, i, p, 5, 5, 5, 0, &h) != RSP_OK) return fail("upload");
    double means[5], rows[5];
    if (rsp_csc_column_sums(h, got) != RSP_OK || memcmp(got, want, sizeof want) != 0) return fail("csc_column_sums");
    if (rsp_csc_column_means(h, means) != RSP_OK) return fail("csc_column_means");
    for (int c = 0; c < 5; ++c) if (means[c] != want[c] / 5) { fprintf(stderr, "means differ\n"); return 1; }
    if (rsp_csc_row_sums(h, rows) != RSP_OK) return fail("csc_row_sums");
    const double want_rows[5] = {0.41 + 0.84, 0.37 + 0.26, 0.35, 0.0, 0.0};
    for (int r = 0; r < 5; ++r) if (fabs(rows[r] - want_rows[r]) > 1e-15) { fprintf(stderr, "rows differ\n"); return 1; }
    rsp_csc_free(h);

    /* error behaviour: invalid offsets are rejected, nothing is computed */
    const int32_t bad[6] = {0, 2, 1, 3, 4, 5};
    if (rsp_column_sums_host(x, bad, 5, 5, got, 0) != RSP_ERR_BAD_ARG) { fprintf(stderr, "bad p accepted\n"); return 1; }

    /* ragged matrix, 3e6 nnz */
    const int ncol = 70001;
    int32_t *pp = malloc(sizeof(int32_t) * (ncol + 1));
    unsigned long long s = 88172645463325252ull;
    pp[0] = 0;
    for (int c = 0; c < ncol; ++c) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        int len = (c % 11 == 0) ? 0 : (int)(s % 90);
        if (c == 1234) len = 250000;
        pp[c + 1] = pp[c] + len;
    }
    const long nnz = pp[ncol];
    double *xx = malloc(sizeof(double) * nnz), *ref = malloc(sizeof(double) * ncol), *out = malloc(sizeof(double) * ncol);
    for (long k = 0; k < nnz; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; xx[k] = (double)((long)(s % 2001) - 1000) / 100.0; }
    for (int c = 0; c < ncol; ++c) { double a = 0.0; for (int j = pp[c]; j < pp[c + 1]; ++j) a += xx[j]; ref[c] = a; }
    if (rsp_column_sums_host(xx, pp, ncol, nnz, out, 0) != RSP_OK) return fail("ragged");
    for (int c = 0; c < ncol; ++c) {
        double l1 = 0.0; for (int j = pp[c]; j < pp[c + 1]; ++j) l1 += fabs(xx[j]);
        if (fabs(out[c] - ref[c]) > 1e-12 * l1) { fprintf(stderr, "column %d: %.17g vs %.17g\n", c, out[c], ref[c]); return 1; }
    }
    /* round 5: the same call again -- the library has kept its stream and buffers; give them back; the next call simply
     * allocates again.  One knob entry instead of a setter per knob; an unknown key is an error, not a crash. */
    if (rsp_column_sums_host(xx, pp, ncol, nnz, out, 0) != RSP_OK) return fail("ragged, second call");
    if (rsp_release_cached() != RSP_OK) return fail("release_cached");
    if (rsp_column_sums_host(x, p, 5, 5, got, 0) != RSP_OK || memcmp(got, want, sizeof want) != 0) return fail("after release_cached");
    int knob = -1;
    if (rsp_debug_get("auto_plan", &knob) != RSP_OK || knob != 1) return fail("debug_get auto_plan");
    if (rsp_debug_set("auto_plan", 0) != RSP_OK || rsp_debug_get("auto_plan", &knob) != RSP_OK || knob != 0) return fail("debug_set");
    if (rsp_debug_set("auto_plan", 1) != RSP_OK) return fail("debug_set back");
    if (rsp_debug_set("no_such_knob", 1) != RSP_ERR_BAD_ARG) { fprintf(stderr, "unknown knob accepted\n"); return 1; }
    if (rsp_column_sums_device_form(NULL, 5, 5, 0) != -1) { fprintf(stderr, "a form for offsets never seen\n"); return 1; }
    /* round 6: the single-process multi-GPU handle -- what an R package's C code holds behind gpuMatrix(A, devices = ...).
     * Three shards on device 0; every launch x gather gives the same bits; the page-locked result vector as destination;
     * rowSums reduced on the device(s); and, with ONE shard, the RCCL gather (ncclCommInitAll) -- in THIS process RCCL is
     * the system's librccl.so, not torch's bundled copy: rsp_rccl_info says which. */
    {
        int32_t *ii = malloc(sizeof(int32_t) * (nnz > 0 ? nnz : 1));
        const int32_t nrow_m = 5000;
        for (int c = 0; c < ncol; ++c) for (int j = pp[c]; j < pp[c + 1]; ++j) ii[j] = (int32_t)(((long)(j - pp[c]) * 7919 + c) % nrow_m);
        const int devs3[3] = {0, 0, 0};
        rsp_mcsc_t m = NULL;
        if (rsp_mcsc_upload_csc(xx, ii, pp, nrow_m, ncol, nnz, devs3, 3, &m) != RSP_OK) return fail("mcsc_upload_csc");
        double *first = malloc(sizeof(double) * ncol);
        if (rsp_mcsc_column_sums(m, first) != RSP_OK) return fail("mcsc_column_sums");
        for (int c = 0; c < ncol; ++c) {
            double l1 = 0.0; for (int j = pp[c]; j < pp[c + 1]; ++j) l1 += fabs(xx[j]);
            if (fabs(first[c] - ref[c]) > 1e-12 * l1) { fprintf(stderr, "mcsc column %d\n", c); return 1; }
        }
        const int gathers[3] = {RSP_GATHER_D2H, RSP_GATHER_BLIT, RSP_GATHER_STORES};
        for (int launch = RSP_LAUNCH_SERIAL; launch <= RSP_LAUNCH_WORKERS; ++launch)
            for (int g = 0; g < 3; ++g) {
                if (rsp_mcsc_set_launch(m, launch) != RSP_OK || rsp_mcsc_set_gather(m, gathers[g]) != RSP_OK) return fail("mcsc_set_*");
                if (rsp_mcsc_column_sums(m, out) != RSP_OK) return fail("mcsc_column_sums (mode)");
                if (memcmp(out, first, sizeof(double) * ncol) != 0) { fprintf(stderr, "mcsc bits differ (launch %d gather %d)\n", launch, gathers[g]); return 1; }
            }
        double *pinned = rsp_mcsc_result_buffer(m);
        if (!pinned || rsp_mcsc_column_sums(m, pinned) != RSP_OK || memcmp(pinned, first, sizeof(double) * ncol) != 0) return fail("mcsc result buffer");
        if (rsp_mcsc_set_gather(m, RSP_GATHER_RCCL) != RSP_ERR_BAD_ARG) { fprintf(stderr, "RCCL gather accepted with two shards on one device\n"); return 1; }
        int32_t cfg[4];
        if (rsp_mcsc_config(m, cfg) != RSP_OK || cfg[2] != 2) { fprintf(stderr, "expected two parked workers, config says %d\n", cfg[2]); return 1; }
        double *rsum = malloc(sizeof(double) * nrow_m), *rref = calloc(nrow_m, sizeof(double)), *rl1 = calloc(nrow_m, sizeof(double));
        for (long k = 0; k < nnz; ++k) { rref[ii[k]] += xx[k]; rl1[ii[k]] += fabs(xx[k]); }
        if (rsp_mcsc_row_sums(m, rsum) != RSP_OK) return fail("mcsc_row_sums");
        for (int r = 0; r < nrow_m; ++r) if (fabs(rsum[r] - rref[r]) > 1e-12 * rl1[r]) { fprintf(stderr, "mcsc row %d: %.17g vs %.17g\n", r, rsum[r], rref[r]); return 1; }
        rsp_mcsc_free(m);
        const int dev1[1] = {0};
        if (rsp_mcsc_upload(xx, pp, nrow_m, ncol, nnz, dev1, 1, &m) != RSP_OK) return fail("mcsc_upload (one shard)");
        if (rsp_mcsc_column_sums(m, first) != RSP_OK) return fail("mcsc one shard");
        if (rsp_mcsc_set_gather(m, RSP_GATHER_RCCL) != RSP_OK) return fail("mcsc_set_gather RCCL (ncclCommInitAll over one device)");
        if (rsp_mcsc_column_sums(m, out) != RSP_OK || memcmp(out, first, sizeof(double) * ncol) != 0) return fail("mcsc RCCL gather");
        rsp_mcsc_free(m);
        int rccl_version = 0;
        char rccl_path[512];
        if (rsp_rccl_info(&rccl_version, rccl_path, sizeof rccl_path) != RSP_OK || rccl_version < 20000) return fail("rccl_info");
        printf("RCCL %d from %s\n", rccl_version, rccl_path);
        free(ii); free(first); free(rsum); free(rref); free(rl1);
    }
    printf("cabi_smoke ok (%ld nnz, %d columns)\n", nnz, ncol);
    return 0;
}

/*
 * nohip_stubs.c -- compiled into the package ONLY when it is built on a machine without
 * librcppsparse_hip.so (./configure decides; see Makevars.in).  Every entry of
 * include/rcppsparse_hip.h that the package's C++ calls exists here with the answer a
 * machine without any GPU gives: rsp_device_count() reports zero devices and every
 * compute entry fails with RSP_ERR_NO_DEVICE.  columnSums() then answers with the
 * reference's own column loop on the host (columnsums_impl.hpp), gpuMatrix() is an R
 * error -- exactly what the HIP build does on a box without a GPU, so the R-visible
 * behaviour of the two builds is the same there.  No GPU code, no fallback arithmetic:
 * nothing in this file computes anything.
 */
#include "../inst/include/rcppsparse_hip.h"

static const char *k_msg = "no HIP device available (this build of RcppSparse was made without librcppsparse_hip)";

const char *rsp_version(void) { return "rcppsparse_hip (not linked: host-only build)"; }
const char *rsp_last_error(void) { return k_msg; }
int rsp_device_count(int *count) {
    if (!count) return RSP_ERR_BAD_ARG;
    *count = 0;
    return RSP_OK;
}
int rsp_column_sums_host(const double *x, const int32_t *p, int32_t ncol, int64_t nnz, double *sums, int device) {
    (void)x; (void)p; (void)ncol; (void)nnz; (void)sums; (void)device;
    return RSP_ERR_NO_DEVICE;
}
int rsp_column_sums_host_multi(const double *x, const int32_t *p, int32_t ncol, int64_t nnz, double *sums, const int *devices,
                               int ndevices) {
    (void)x; (void)p; (void)ncol; (void)nnz; (void)sums; (void)devices; (void)ndevices;
    return RSP_ERR_NO_DEVICE;
}
int rsp_release_cached(void) { return RSP_OK; }   /* nothing is ever kept in this build */
int rsp_csc_upload(const double *x, const int32_t *i, const int32_t *p, int32_t nrow, int32_t ncol, int64_t nnz,
                   int device, rsp_csc_t *handle) {
    (void)x; (void)i; (void)p; (void)nrow; (void)ncol; (void)nnz; (void)device;
    if (handle) *handle = 0;
    return RSP_ERR_NO_DEVICE;
}
int rsp_mcsc_upload_csc(const double *x, const int32_t *i, const int32_t *p, int32_t nrow, int32_t ncol, int64_t nnz,
                        const int *devices, int ndevices, rsp_mcsc_t *handle) {
    (void)x; (void)i; (void)p; (void)nrow; (void)ncol; (void)nnz; (void)devices; (void)ndevices;
    if (handle) *handle = 0;
    return RSP_ERR_NO_DEVICE;
}
/* (no handle can exist in this build: the entries below are only ever reached with NULL) */
int rsp_csc_dims(rsp_csc_t h, int32_t *nrow, int32_t *ncol, int64_t *nnz) { (void)h; (void)nrow; (void)ncol; (void)nnz; return RSP_ERR_BAD_ARG; }
int rsp_csc_column_sums(rsp_csc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_csc_column_means(rsp_csc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_csc_row_sums(rsp_csc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_csc_row_means(rsp_csc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_csc_crossprod(rsp_csc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_csc_free(rsp_csc_t h) { (void)h; return RSP_OK; }
int rsp_mcsc_dims(rsp_mcsc_t h, int32_t *nrow, int32_t *ncol, int32_t *nshards) { (void)h; (void)nrow; (void)ncol; (void)nshards; return RSP_ERR_BAD_ARG; }
int rsp_mcsc_column_sums(rsp_mcsc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_mcsc_column_means(rsp_mcsc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_mcsc_row_sums(rsp_mcsc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_mcsc_row_means(rsp_mcsc_t h, double *out) { (void)h; (void)out; return RSP_ERR_NO_DEVICE; }
int rsp_mcsc_free(rsp_mcsc_t h) { (void)h; return RSP_OK; }

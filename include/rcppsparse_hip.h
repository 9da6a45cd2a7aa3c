/*
 * rcppsparse_hip.h -- C ABI of librcppsparse_hip.so (MI355X / gfx950).
 *
 * This is the device boundary inserted *inside* the reference's columnSums:
 * everything above it (R wrapper, Rcpp glue, RcppSparse::Matrix, the Exporter)
 * keeps its shape; the double loop of reference src/example.cpp:28-30, which
 * drives Matrix::InnerIterator (reference inst/include/RcppSparse.h:218-233)
 * over the dgCMatrix slots x / p, is replaced by one call below.
 *
 * Rules of the boundary:
 *   - plain C types only; no Rcpp / R / torch types; no exceptions cross it;
 *   - every entry point returns an int status (RSP_OK == 0); the text of the
 *     last failure on the calling thread is rsp_last_error();
 *   - host pointers are *borrowed* for the duration of one call and never
 *     retained or freed; handles own device copies only;
 *   - output buffers are allocated by the caller (the Rcpp side allocates the
 *     NumericVector on the R main thread before calling in);
 *   - no entry point calls any R API; all are safe to call sequentially from
 *     one thread; they return only when `out` is completely written (host
 *     variants) or when the work is enqueued on `stream` (device variants);
 *   - there is NO CPU fallback in this library: with no usable HIP device the
 *     compute entry points fail with RSP_ERR_NO_DEVICE.
 *
 * Index types follow the reference: p[] and i[] are 32-bit int
 * (RcppSparse.h:30, :232), so nnz <= 2^31-1; byte offsets are 64-bit inside.
 * ncol may be up to 2^31 - 65537 (more is RSP_ERR_BAD_ARG).
 * Contract on p (what Matrix::dgCMatrix guarantees and the reference assumes
 * without checking): p[0] == 0, p non-decreasing, p[ncol] == nnz.  The host
 * entry points verify this and return RSP_ERR_BAD_ARG otherwise; the device
 * entry points trust the caller (their reads and writes stay in bounds for
 * any p -- tests/test_gpu_parity.py::test_invalid_offsets_on_a_device_entry_stay_in_bounds --
 * but the sums are then unspecified).
 */
#ifndef RCPPSPARSE_HIP_H
#define RCPPSPARSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSP_OK              0
#define RSP_ERR_NO_DEVICE   1   /* no HIP device / runtime unusable            */
#define RSP_ERR_BAD_ARG     2   /* null pointer, negative size, invalid p[]    */
#define RSP_ERR_HIP         3   /* a HIP runtime call failed                   */
#define RSP_ERR_WORKSPACE   4   /* workspace too small                         */
#define RSP_ERR_RCCL        5   /* an RCCL call failed                         */
#define RSP_ERR_ALLOC       6   /* host or device allocation failed            */

/* ---- library / device queries ------------------------------------------ */

/* "rcppsparse_hip <semver> gfx950". */
const char *rsp_version(void);
/* Message of the last non-OK status returned on this thread ("" if none). */
const char *rsp_last_error(void);
/* Number of visible HIP devices; 0 (and RSP_OK) when there are none -- and in a process that was FORKED from one that had
 * already used the GPU through this library (R's parallel::mclapply): the HIP runtime does not survive a fork, so such a
 * child is a machine without a device (the host entries answer RSP_ERR_NO_DEVICE without touching the runtime; the Rcpp
 * layer above the ABI then runs the reference's loop on the host). */
int rsp_device_count(int *count);

/* ---- one-shot path: replaces reference src/example.cpp:28-30 ------------ */
/*
 * sums[c] = sum_{j = p[c]}^{p[c+1]-1} x[j]   for c in [0, ncol)
 * x: REAL(A@x), p: INTEGER(A@p) (host memory, borrowed), ncol = A@Dim[1]
 * (RcppSparse.h:45 cols()), nnz = length(A@x) (RcppSparse.h:48 n_nonzero()).
 * `sums` is the caller-allocated NumericVector storage of example.cpp:27 (it
 * need not be zero-filled).  Uploads x and p to `device`, runs the segmented
 * sum there, copies the ncol doubles back.  i[] is not needed: the reference
 * never reads it on this path (RcppSparse.h:227 row() is not called).
 */
int rsp_column_sums_host(const double *x, const int32_t *p, int32_t ncol,
                         int64_t nnz, double *sums, int device);
/*
 * What the library keeps between calls, and how to give it back.  rsp_column_sums_host keeps ONE stream and
 * one grow-only set of device buffers (x, p, sums, workspace) per device, so that an R session calling
 * columnSums(A) again and again does not pay a stream creation, four allocations and four frees per call (they
 * cost more than the transfers and kernels of any matrix below ~1e7 entries).  A call that would need more than
 * 1 GiB in total (RSP_ONE_SHOT_KEEP_MB in the environment) allocates for itself and frees before it returns.
 * Host pointers are never kept.  Concurrent one-shot calls on one device take turns (a mutex per device);
 * calls on different devices run side by side.  rsp_release_cached() frees everything kept, on every device
 * (the R package calls it when it is unloaded); the next call simply allocates again.
 */
int rsp_release_cached(void);

/*
 * The same one-shot call spread over several GPUs of the node: the columns are cut
 * into nnz-balanced contiguous ranges, every range is uploaded over its own GPU's
 * host link by its own host thread, summed there, and its slice of the sums is
 * copied straight into `sums` (no collective: the result lives in host memory).
 * `devices` lists one device ordinal per shard (an ordinal may repeat); NULL / 0
 * means one shard on every visible device.  Same contract on x, p, sums as above.
 */
int rsp_column_sums_host_multi(const double *x, const int32_t *p, int32_t ncol,
                               int64_t nnz, double *sums, const int *devices,
                               int ndevices);

/*
 * Upload once, sum many, over several GPUs from ONE process -- the multi-GPU path an R session reaches (the reference's
 * .Call runs on the R main thread, src/RcppExports.cpp:16-24): the columns are cut as above (SURVEY.md 8e:
 * bounds[k] = lower_bound(p, k * nnz / G)), every range becomes a resident shard on its device, and
 * rsp_mcsc_column_sums runs all shards side by side and lands every slice in `sums`.  `devices` as for
 * rsp_column_sums_host_multi.
 *
 * A call creates NOTHING (round 6; before, G - 1 threads were started and joined per call): the handle keeps per shard
 * a stream, an output, a plan and an event, and per handle one page-locked vector of ncol doubles, parked worker threads and,
 * for the RCCL gather, the communicators of one ncclCommInitAll.  Two independent choices, both per handle:
 *   launch  RSP_LAUNCH_SERIAL   the calling thread enqueues every shard's kernels, then every shard's trip home, then
 *                               polls the shards' events in turn (a shard is handed over when IT is done);
 *           RSP_LAUNCH_WORKERS  shard 0 on the calling thread, shard k on its own thread, which stays parked between
 *                               calls (futex; it spins for RSP_MCSC_SPIN_US microseconds, default 50, after a call so
 *                               that calls in a loop find it awake): enqueues, waits and the copies into `sums` overlap.
 *                               Default from 3 shards on (RSP_MCSC_LAUNCH=serial|workers in the environment overrides).
 *   gather  RSP_GATHER_D2H      every shard copies its slice over ITS device's host link into the page-locked vector
 *                               (hipMemcpyAsync; the default where shards share a device);
 *           RSP_GATHER_BLIT     the same trip by a copy kernel of the library behind the shard's kernels: it starts a few
 *                               microseconds after them, where the runtime's copy command needs ~20 us before its first
 *                               byte moves (1 MB slice: 19 against 36 us).  The default with a device per shard
 *                               (RSP_MCSC_GATHER=d2h|blit in the environment overrides);
 *           RSP_GATHER_RCCL     the slices travel to shard 0's device in one group of ncclSend / ncclRecv over xGMI
 *                               (communicators from ncclCommInitAll, made when the mode is first selected; one DEVICE
 *                               per shard, RCCL refuses duplicates), then ONE copy of the whole vector to the host;
 *           RSP_GATHER_STORES   the kernels take the page-locked vector as their output: no copy command at all.
 * Every combination returns the bits of the per-shard device calls (the same launches over the same data).
 * `sums` may be rsp_mcsc_result_buffer(handle) -- the page-locked vector itself, ncol doubles, valid until the next
 * call on the handle or its release: then nothing is copied on the host at all.
 * fork(): a child process (R's parallel::mclapply) inherits the handle's pointer but neither its worker threads nor a
 * usable GPU context; the handle's entries fail there with RSP_ERR_BAD_ARG and a message (they do not hang), and
 * rsp_mcsc_free in the child releases nothing of the parent's.
 */
typedef struct rsp_mcsc *rsp_mcsc_t;
#define RSP_GATHER_D2H     0
#define RSP_GATHER_RCCL    1
#define RSP_GATHER_STORES  2
#define RSP_GATHER_BLIT    4   /* a copy KERNEL of the library writes the slice into the page-locked vector behind the shard's
                                  kernels (instead of the runtime's copy command, which needs ~20 us before its first byte moves) */
#define RSP_GATHER_NONE    3   /* MEASUREMENT ONLY: the slices stay on the devices, `sums` is not written; the call
                                  returns when every shard's stream has drained (launch + wait cost without a transfer) */
#define RSP_LAUNCH_SERIAL  0
#define RSP_LAUNCH_WORKERS 1
int rsp_mcsc_upload(const double *x, const int32_t *p, int32_t nrow, int32_t ncol,
                    int64_t nnz, const int *devices, int ndevices, rsp_mcsc_t *handle);
int rsp_mcsc_column_sums(rsp_mcsc_t handle, double *sums);
/* Matrix::colMeans (RcppSparse.h:145-150) on the shards: sums divided by Dim[0]. */
int rsp_mcsc_column_means(rsp_mcsc_t handle, double *means);
int rsp_mcsc_set_gather(rsp_mcsc_t handle, int mode);
int rsp_mcsc_set_launch(rsp_mcsc_t handle, int mode);
/* info4 = { gather, launch, worker threads alive, communicators made } */
int rsp_mcsc_config(rsp_mcsc_t handle, int32_t *info4);
double *rsp_mcsc_result_buffer(rsp_mcsc_t handle);
/*
 * The same handle over shards that ALREADY live in the devices' HBM (a caller that produced x / p on the GPUs, bench.py):
 * shard k = columns [sum of shard_ncol[0..k), ...) with its entries d_x[k] (16-byte aligned), its rebased offsets
 * d_p[k] (shard_ncol[k] + 1 values from 0 to shard_nnz[k]) and optionally its row indices d_i[k] (d_i may be NULL), all
 * on devices[k].  Nothing is copied and nothing of the caller's is ever freed; the memory must outlive the handle.
 * Every shard's offsets are inspected on its device (rsp_column_sums_plan_create_device, waited for), so calls take
 * their final form from the first one on.
 */
int rsp_mcsc_wrap_device(int nshards, const int *devices, const double *const *d_x,
                         const int32_t *const *d_i, const int32_t *const *d_p,
                         const int32_t *shard_ncol, const int64_t *shard_nnz, int32_t nrow,
                         rsp_mcsc_t *handle);
/* Measurement: host clock of the LAST column-sum call, microseconds from its entry: us[0] = the whole call, then per
 * shard { enqueue begun, enqueue returned, stream drained, slice copied out }; capacity >= 1 + 4 * shards. */
int rsp_mcsc_last_call_stamps(rsp_mcsc_t handle, double *us, int capacity);
/* Measurement: mean device milliseconds of ONE shard's column-sum launches alone (HIP events on its stream). */
int rsp_mcsc_shard_kernel_ms(rsp_mcsc_t handle, int32_t shard, int reps, float *ms);
/* The same handle with the row indices kept on the devices (i[] cut like x[]), for the row-wise entries:
 * Matrix::rowSums / rowMeans (RcppSparse.h:138-156).  Every shard sums the rows of its own columns into a vector in
 * its HBM; the vectors are added in shard order = column order ON THE DEVICES (round 6): the rows are cut into one
 * slice per shard, device r copies slice r of every other shard's vector to itself (hipMemcpyPeerAsync, every pair of
 * devices its own xGMI link), adds the pieces in shard order with one kernel and copies its reduced slice home over
 * its own host link -- the single-process form of rsp_comm_reduce_rows, same sum term for term.  Per device this keeps
 * nrow + (shards + 1) * nrow / shards doubles of HBM from the first call on.  If that memory cannot be had (or
 * RSP_MCSC_ROWS=host in the environment) the vectors come back over the host links and the host adds them: the same
 * bits, nshards * nrow doubles of host memory for the duration of the call.  `sums` / `means`: nrow doubles, host. */
int rsp_mcsc_upload_csc(const double *x, const int32_t *i, const int32_t *p, int32_t nrow,
                        int32_t ncol, int64_t nnz, const int *devices, int ndevices,
                        rsp_mcsc_t *handle);
int rsp_mcsc_row_sums(rsp_mcsc_t handle, double *sums);
int rsp_mcsc_row_means(rsp_mcsc_t handle, double *means);
/* Dim[0], Dim[1] and the number of shards, from the handle itself (any output may be NULL). */
int rsp_mcsc_dims(rsp_mcsc_t handle, int32_t *nrow, int32_t *ncol, int32_t *nshards);
/* info4 = { first column, one past the last column, column-sum form (rsp_csc_column_form), entries } of a shard */
int rsp_mcsc_shard_info(rsp_mcsc_t handle, int32_t shard, int32_t *info4);
/* Threads: calls on one handle must not overlap
 * (as for rsp_csc_t); different handles may be used from different threads. */
int rsp_mcsc_free(rsp_mcsc_t handle);

/* ---- device-resident dgCMatrix handle (upload once, sum many) ---------- */
/* The slot layout x / i / p / Dim of reference RcppSparse.h:29-30 is the wire
 * format; i may be NULL (it is only kept for the row-wise "next" entries).
 * A handle is owned by one thread at a time (calls on the same handle must not
 * overlap); different handles may be used from different threads.  The handle
 * entries (and the multi-GPU host entries) run on the handle's device and put the
 * calling thread's current HIP device back before they return. */
typedef struct rsp_csc *rsp_csc_t;

int rsp_csc_upload(const double *x, const int32_t *i, const int32_t *p,
                   int32_t nrow, int32_t ncol, int64_t nnz, int device,
                   rsp_csc_t *handle);
/* columnSums on the resident copy; `sums` is host memory, ncol doubles. */
int rsp_csc_column_sums(rsp_csc_t handle, double *sums);
/* Matrix::colMeans (RcppSparse.h:145-150): column sums divided by Dim[0]. */
int rsp_csc_column_means(rsp_csc_t handle, double *means);
/* Dim[0], Dim[1], length(x) of the resident copy (any output may be NULL). */
int rsp_csc_dims(rsp_csc_t handle, int32_t *nrow, int32_t *ncol, int64_t *nnz);
/* The upload inspects p[] once -- on a host thread beside the copies, or (65536 columns and more; round 5) on the
 * device behind the copy of p[], where it costs microseconds instead of milliseconds -- and freezes the result in the
 * handle.  The two inspectors select the same form: where the device-made lean image (sized before the offsets are
 * seen) has no room for a matrix's densest chunk, the upload inspects that matrix again on the host (round 6), so a
 * matrix of short columns gets the lean form -- every column bit-identical to the reference loop -- on either side of
 * 65536 columns.  (rsp_mcsc_wrap_device, whose offsets never visit the host, keeps the device-made plan.)  Which form the
 * handle's column sums take -- 0 general kernels, 1 snapped, 2 lean, 3 columns (rsp_column_sums_plan_info).
 * The settings in force AT UPLOAD decide (RSP_LEAN, RSP_COLUMNS_FORM, the chunking knobs: rsp_debug_set);
 * changing them later does not touch existing handles.  rsp_csc_set_planned(h, 0) sends the handle's
 * column sums through the general kernels whatever the plan says (A/B measurements), 1 switches back. */
int rsp_csc_column_form(rsp_csc_t handle);
int rsp_csc_set_planned(rsp_csc_t handle, int on);
int rsp_csc_free(rsp_csc_t handle);

/* ---- device-pointer path (inputs already in HBM) ----------------------- */
/*
 * Same computation on device pointers; everything is enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream) of the calling
 * thread's current HIP device, and nothing synchronises.  d_x must be 16-byte
 * aligned.  d_workspace is scratch of at least
 * rsp_column_sums_workspace_bytes(ncol, nnz) bytes, 16-byte aligned; it carries
 * no state between calls, but calls that may run concurrently (different streams)
 * need a workspace each.  Graph-capture safe (a call on a capturing stream
 * allocates nothing and waits for nothing).  Many calls on small or medium
 * matrices: alternating them over two
 * streams (two workspaces) lets one call fill the chip while the previous one
 * drains (bench.py does this for the 1/8 shards of the multi-GPU runs: -15 %).
 *
 * These two entries PLAN FOR THEMSELVES (matrices of at least 2^20 entries; round 6 rules).  The FIRST call on
 * (device, d_p, ncol, nnz) runs the general kernels and only notes the key (no allocation, nothing else enqueued: a
 * caller that shows fresh offsets in every call never pays for a plan).  The SECOND call runs the general kernels too
 * and enqueues a device-side inspection of d_p behind them on `stream` (rsp_column_sums_plan_create_device: ~23 us for
 * 1e6 columns; its memory is allocated in this call, once per key).  Once the host has seen the result -- one load of a
 * page-locked word per call: no event query, never a wait -- later calls with the same key take the form it selects:
 * lean (every column short: ONE launch, every column bit-identical to the reference loop; BASELINE config 2: 0.51 ->
 * ~0.7 of the HBM roofline) or columns (every column long: one launch); other matrices stay on the general kernels.
 * The caller promises NOTHING about d_p between calls: the kernels of this path check every column's offsets against
 * the p[] of the call they run in, sum a column whose offsets have changed straight from x (clamped to [0, nnz]), and
 * make the library retire that plan and inspect again -- never a wrong sum, only a slower call.
 * NOTHING IN THESE ENTRIES WAITS FOR THE DEVICE.  A retired plan's image is given up when an event recorded on every
 * stream it was launched on has completed (looked at only while something is retired) -- given up, not freed: on this
 * runtime hipFree and hipHostFree drain every stream of the device first, so the allocation goes to a pool of at most 8
 * (and at most 1 GiB in all) and the next plan that fits takes it (a re-inspection of the same key always fits; only what
 * overflows the pool -- dead images of sizes nobody asks for again -- is really freed, and that free waits).  A key keeps at most 2 retired images (beyond that it
 * stays on the general kernels until one is given up), so HBM use is bounded whatever the caller does with d_p.  Up to 16 keys are remembered per process; keys that never got a plan make room first, a planned key
 * only after 64 calls without a use (its plan is retired, not waited for).  One plan is launched on up to 6 different
 * streams; calls on further streams take the general kernels.  rsp_release_cached() forgets everything (it DOES wait).
 * BIT STABILITY: every form is deterministic, but the general kernels and a planned form agree within
 * the documented tolerance, not bit for bit, and WHICH call is the first planned one depends on when the inspection's
 * result is seen.  rsp_column_sums_device_settle(d_p, ncol, nnz, stream) removes that: it makes the key's plan now if
 * there is none (inspection enqueued on `stream`), WAITS for its result and returns the form -- 0 general kernels,
 * 2 lean, 3 columns, -1 no usable device.  From its return on, every call with that key takes that one form and
 * returns identical bits run to run, for as long as d_p holds the same offsets and the key stays among the remembered
 * ones (tests/test_gpu_autoplan.py::test_bit_stable_run_to_run_under_the_defaults).  Without it the guarantee starts
 * at the first call after rsp_column_sums_device_form(.., wait = 1) has returned >= 0.
 * A call on a CAPTURING stream always records the general kernels (a graph outlives the call, the library's plan images
 * do not belong to it): to put the planned form into a graph, make the plan yourself
 * (rsp_column_sums_plan_create_device, rsp_column_sums_plan_wait) and capture rsp_column_sums_planned_device -- the
 * plan's lifetime is then yours.  RSP_AUTO_PLAN=0 in the environment (or rsp_debug_set("auto_plan", 0)) keeps every
 * call on the general kernels, bit-stable from the first call; explicit plans (below) and handles choose their form
 * once, at creation / upload.
 * rsp_column_sums_device_form: the form calls with that key take now -- 0 general kernels, 2 lean, 3 columns,
 * -1 not known (yet: no plan has been made, or its result has not been seen); wait != 0 blocks until the result of an
 * inspection that has been enqueued is seen (it does not make a plan: that is settle's job).
 */
size_t rsp_column_sums_workspace_bytes(int32_t ncol, int64_t nnz);
int rsp_column_sums_device(const double *d_x, const int32_t *d_p, int32_t ncol,
                           int64_t nnz, double *d_sums, void *d_workspace,
                           size_t workspace_bytes, void *stream);
/* As above, then sums[c] /= nrow (RcppSparse.h:145-150), fused in the same
 * launches. */
int rsp_column_means_device(const double *d_x, const int32_t *d_p, int32_t nrow,
                            int32_t ncol, int64_t nnz, double *d_means,
                            void *d_workspace, size_t workspace_bytes,
                            void *stream);
int rsp_column_sums_device_form(const int32_t *d_p, int32_t ncol, int64_t nnz, int wait);
int rsp_column_sums_device_settle(const int32_t *d_p, int32_t ncol, int64_t nnz, void *stream);
/*
 * Inspector-executor form for callers that can show p[] to the host once (a resident matrix summed many
 * times; rsp_csc_upload does this by itself, the one-shot rsp_column_sums_host does not: it sums once).  The inspector walks the chunk
 * grid over p[] (reference RcppSparse.h:220-221: a column is [p[c], p[c+1])) and records for every chunk the
 * first column that starts in it, with chunk boundaries snapped to that column start.  If no column reaches
 * more than one group (512 entries) past a chunk edge, a planned call is ONE launch: no per-chunk column search,
 * no carries, no fix-up launch, no workspace -- what a latency-bound call (BASELINE config 2: 1e7 entries)
 * spends a third of its time on.  Otherwise the plan is marked "not snapped" and the executor runs the general
 * kernels (it then needs the workspace of rsp_column_sums_workspace_bytes).  Results are those of
 * rsp_column_sums_device within the same tolerance; the short-column paths are the same code, so columns of up
 * to 16 entries inside a group stay bit-identical to the reference loop.  A plan belongs to the p[] it was made
 * from (same ncol, nnz, offsets), to the chunking in force when it was made and to the device it was made on
 * (a planned call from a thread whose current device is another one is RSP_ERR_BAD_ARG).
 * When in addition every column is short (at most 64 entries; BASELINE config 2) the plan takes the LEAN form:
 * the inspector rewrites the offsets a chunk needs as 16-bit column starts relative to the chunk, at a fixed
 * stride, so that a wavefront requests its rows of x, its header and its offsets in the same instant (2 B per
 * column, p[] itself is not read again), puts both into LDS and adds every column in storage order from +0.0:
 * every column then comes out BIT-IDENTICAL to the reference loop.  It is selected up to a mean column length of 60
 * (longer columns leave a chunk's 64 lanes too few columns: tools/edge_sweep.py); RSP_LEAN=0 (rsp_debug_set("lean", 0)) keeps
 * plans out of that form, 2 takes it wherever every column is <= 64 entries (A/B measurements, tests).
 * When every column is LONG and of similar length (at least 2048 entries -- 512 in matrices of up to 2.5e8
 * entries --, none above four times the mean, at least 128 columns; the reference vignette's 100000 x 1000 benchmark matrix) the plan takes the COLUMNS form:
 * nothing is recorded at all, a call is one launch of one workgroup per column that reads p[c], p[c + 1] itself
 * (no column search per chunk, no carries, no fix-up launch, no workspace); results within the usual tolerance.
 * RSP_COLUMNS_FORM=0 keeps plans out of it.
 * info4 = { form (0 general kernels, 1 snapped, 2 lean, 3 columns), chunks (columns form: columns), entries per
 * chunk (columns form: threads per column), largest distance from a chunk's grid start to its first column start
 * (lean: most columns in one chunk; columns form: the longest column) }; *inspect_ms = host time the
 * inspection took (reported separately from the calls).
 * nrow_for_means > 0: colMeans (RcppSparse.h:145-150), 0: sums.
 */
typedef struct rsp_colsums_plan *rsp_colsums_plan_t;
int rsp_column_sums_plan_create(const int32_t *p, int32_t ncol, int64_t nnz, int device,
                                rsp_colsums_plan_t *plan);
/*
 * The same for offsets that live in HBM, WITHOUT showing them to the host: the inspection runs as three small
 * kernels on `stream` (one pass over p[] with a thread per column: validity of p[], column lengths, the snapped
 * records, the lean grid's first columns; then the lean chunks' widths; then the lean image), a few tens of
 * microseconds for 1e6 columns, and nothing synchronises.  The images equal those of rsp_column_sums_plan_create
 * bit for bit (tests/test_gpu_parity.py); the few statistics the choice of form rests on travel to a page-locked
 * host record behind the kernels.  Until the host has seen them, rsp_column_sums_planned_device answers with the
 * general kernels (right for any matrix; pass their workspace), from then on with the form they select -- it looks
 * (hipEventQuery, no waiting) at every call, so a caller that plans and then sums in a loop gets the planned form
 * from the second call or so on, and never blocks.  rsp_column_sums_plan_ready looks without calling;
 * rsp_column_sums_plan_wait blocks until the inspection is done (a caller about to CAPTURE planned calls into a
 * HIP graph waits first: a capture records whatever form is known at that moment, and looks at nothing itself);
 * rsp_column_sums_plan_info waits too, and its *inspect_ms is then the device time of the inspection.
 * Offsets that are not a dgCMatrix's (p[0] != 0, decreasing, p[ncol] != nnz) are noticed by the inspection: such a
 * plan stays on the general kernels.  Differences from the host-made plan: the lean image is sized before the
 * offsets are seen, for 3 x the mean number of columns per chunk + 16 (at least 126): a matrix with a denser
 * chunk takes the snapped or general form here.  The plan owns its device memory; destroy it after the stream's work.
 */
int rsp_column_sums_plan_create_device(const int32_t *d_p, int32_t ncol, int64_t nnz,
                                       void *stream, rsp_colsums_plan_t *plan);
int rsp_column_sums_plan_ready(rsp_colsums_plan_t plan);   /* 1: the form is known (always, for host-made plans), 0: not yet */
int rsp_column_sums_plan_wait(rsp_colsums_plan_t plan);
/* Test helper: copies a plan's image to the host -- what = 0: the snapped form's (chunks + 1) records, 1: the lean
 * form's headers and 16-bit offsets.  *bytes = its size (0: this plan has no such image); host may be NULL. */
int rsp_debug_plan_image(rsp_colsums_plan_t plan, int what, void *host, size_t capacity, size_t *bytes);
int rsp_column_sums_plan_info(rsp_colsums_plan_t plan, int32_t *info4, double *inspect_ms);
/* ncol, nnz: the sizes of the matrix behind d_x / d_p; they must be the plan's (RSP_ERR_BAD_ARG otherwise:
 * the lean form never reads d_p and would run over a shorter x).  The offsets themselves are the caller's
 * word: a plan belongs to the p[] it was made from. */
int rsp_column_sums_planned_device(rsp_colsums_plan_t plan, const double *d_x,
                                   const int32_t *d_p, int32_t ncol, int64_t nnz,
                                   int32_t nrow_for_means, double *d_sums,
                                   void *d_workspace, size_t workspace_bytes, void *stream);
int rsp_column_sums_plan_destroy(rsp_colsums_plan_t plan);
/*
 * Generic column reduction ("next" row f3): the same column-iteration loop with a
 * different per-element body, out[c] = sum_j f(x[j]) over column c's stored entries --
 * what a user writes with Matrix::InnerIterator for column norms
 * (for (InnerIterator it(A, c); it; ++it) acc += f(it.value());).
 */
#define RSP_OP_SUM          0   /* f(v) = v      (== rsp_column_sums_device) */
#define RSP_OP_SUM_SQUARES  1   /* f(v) = v * v  (squared column 2-norms)     */
#define RSP_OP_SUM_ABS      2   /* f(v) = |v|    (column 1-norms)             */
#define RSP_OP_MAX          3   /* largest stored entry; empty column -> -Inf; NaN entries skipped
                                   (the loop  if (v > acc) acc = v;  from acc = -Inf)          */
#define RSP_OP_MIN          4   /* smallest stored entry; empty column -> +Inf                 */
#define RSP_OP_COUNT        5   /* number of stored entries (Matrix::InnerNNZs, RcppSparse.h:357-359) */
int rsp_column_reduce_device(const double *d_x, const int32_t *d_p, int32_t ncol,
                             int64_t nnz, int op, double *d_out, void *d_workspace,
                             size_t workspace_bytes, void *stream);
/*
 * Row-restricted column sums ("next" row f4): what a loop over
 * Matrix::InnerIteratorInRange (complement = 0) or InnerIteratorNotInRange
 * (complement = 1) computes (reference RcppSparse.h:238-321; documented intent,
 * not its out-of-bounds quirks): out[c] = sum of x[j] over the entries of column c
 * whose row i[j] is / is not in the row set.  The set is a bitmap of nrow bits
 * (row r = bit r % 32 of word r / 32; (nrow + 31) / 32 words in HBM).  Streams
 * x and i (12 B/nnz); d_i must be 8-byte aligned.
 *
 * Forms: up to 2^20 rows the bitmap is probed in L1 / LDS by the column-sum kernel.
 * Above that, when the columns are long enough (nnz >= 32 * ncol * ceil(nrow / 2^20),
 * ncol >= 16384, and a group of ncol / 256..512 columns holds >= 32768 entries per slice),
 * a slice-major kernel walks every column once per slice of 2^20 rows
 * with that slice of the bitmap in LDS; it relies on the rows of a column ascending
 * (dgCMatrix validity; the reference's restricted iterators merge on the same
 * assumption) and needs the workspace of rsp_column_sums_in_rows_workspace_bytes.
 * A device-side check hands matrices with giant columns back to the general kernel
 * (probes served by L2), which is also what a workspace of only
 * rsp_column_sums_workspace_bytes or RSP_ROW_SLICES=0 (rsp_debug_set("row_slices", 0)) select
 * (2: the slice form wherever nrow > 2^20, whatever the shape -- tests).
 * Both forms are deterministic and within 1e-12 * sum|x_col| of the reference's order.
 */
size_t rsp_column_sums_in_rows_workspace_bytes(int32_t nrow, int32_t ncol, int64_t nnz);
/* which form a call of these sizes with that much workspace takes (the slice form's device-side check aside) */
#define RSP_IN_ROWS_FORM_L1     0   /* bitmap <= 16 KB: probed through L1                    */
#define RSP_IN_ROWS_FORM_LDS    1   /* <= 128 KB: whole bitmap in LDS                        */
#define RSP_IN_ROWS_FORM_L2     2   /* larger: general kernel, probes served by L2           */
#define RSP_IN_ROWS_FORM_SLICES 3   /* larger, long columns: slice-major, bitmap slice in LDS */
int rsp_column_sums_in_rows_form(int32_t nrow, int32_t ncol, int64_t nnz, size_t workspace_bytes);
int rsp_column_sums_in_rows_device(const double *d_x, const int32_t *d_i,
                                   const int32_t *d_p, int32_t nrow, int32_t ncol,
                                   int64_t nnz, const uint32_t *d_row_bitmap,
                                   int complement, double *d_out, void *d_workspace,
                                   size_t workspace_bytes, void *stream);
/*
 * Measurement helper: enqueue `reps` back-to-back rsp_column_sums_device calls
 * on `stream`, bracketed by hipEvents recorded on that same stream, wait for
 * the last, and return the mean milliseconds per call in *ms_per_call.
 */
int rsp_column_sums_device_timed(const double *d_x, const int32_t *d_p,
                                 int32_t ncol, int64_t nnz, double *d_sums,
                                 void *d_workspace, size_t workspace_bytes,
                                 void *stream, int reps, float *ms_per_call);

/*
 * Measurement helper ("also record a measured read-only bandwidth on the box as the
 * practical ceiling"): a kernel with the ACCESS SHAPE of rsp_column_sums_device over the same x -- the same
 * chunk grid, one wavefront per chunk, the same 1 KiB nt loads in the same register pipeline, two adds per
 * lane and row -- and none of its column work: p[] is not read, nothing is stored (d_sink: one double, only
 * there so that the loads cannot be optimised away).  The reference reads x once (RcppSparse.h:226); this is
 * how fast this device lets that be done.  One untimed launch, then `reps` launches between two HIP events
 * on `stream`; returns the mean milliseconds per launch (8 * nnz bytes each).  Synchronises `stream`.
 */
/* Test helper: the hand-written exclusive prefix sum of 32-bit counts the regrouping passes use on their count tables
 * (csrc/scan.hip): d_out[k] = d_in[0] + ... + d_in[k - 1]; d_out may be d_in.  Allocates its scratch, waits for `stream`. */
int rsp_debug_exclusive_scan_device(const int32_t *d_in, int32_t *d_out, int64_t n, void *stream);
int rsp_debug_read_ceiling_device(const double *d_x, int64_t nnz, double *d_sink, void *stream,
                                  int reps, float *ms_per_launch);

/* ---- row-wise "next" entries: Matrix::rowSums / rowMeans ----------------- */
/*
 * Reference RcppSparse.h:138-144 / :151-156: sums[i[j]] += x[j] over all stored
 * entries (rowMeans divides by Dim[1]).  Deterministic, no float atomics in
 * global memory, bit-stable run to run, within 1e-12 * sum|x| of the reference's
 * order.  All forms accumulate in LDS, 16384 rows per workgroup, and are hand-written
 * (no library sort): matrices of up to 65536 rows are summed straight from x / i
 * (workspace: the workgroups' partial sums, at most a few hundred MB); larger ones are
 * first regrouped by block of 16384 rows in ONE partition pass (up to 1.36e7 rows), by
 * coarse block of 2 / 4 / 8 such blocks (up to 1.09e8 rows; every row block then picks
 * its entries out of its coarse block's), or in two passes (more rows still), with a
 * workspace of 10 B/nnz where the row blocks themselves are the regrouping's buckets (up to 1.36e7 rows: the copy keeps a
 * row as its 16 bits inside the block; round 6), else 12 B/nnz (two passes: 24 B/nnz; up to 6 % more where the regrouped copy
 * is padded to whole 16-entry groups) + up to 64 B/row + a count table of at most 64 MB.  The handle variants (the handle must have been uploaded with i[]) build
 * that regrouped copy on first use and keep it: repeated calls only accumulate
 * (12 B/nnz); the device variants regroup in the caller's workspace on every call.
 * Entries whose row index is outside [0, nrow) are left out, not added elsewhere.
 * Ask for the workspace size with the device current that will run the call
 * (the plan looks at its CU count).
 * rsp_row_sums_workspace_bytes: 0 = error (sizes out of range).
 *
 * Segments form (handles only -- it needs p[]): where the columns are long (a column has
 * >= 128 entries per block of 16384 rows on average, more than 16384 rows, >= 30 columns)
 * and the rows of every column ascend (dgCMatrix validity, checked once per handle on the
 * device), nothing is regrouped: a table of every column's contiguous piece per row block
 * is built on first use (4 B per column and block) and every call reads the uploaded x / i
 * once, 12 B/nnz, also where the other forms would read them once per row block (2-4
 * blocks) or keep a 12 B/nnz copy.  RSP_ROW_SEGMENTS=0 (rsp_debug_set("row_segments", 0)) keeps
 * handles on the other forms; 2 takes the segments form wherever it is possible (tests).
 * rsp_csc_row_form tells which form a handle's row sums have taken.
 */
#define RSP_ROW_FORM_NONE      0   /* not built yet (no row sums asked for so far)      */
#define RSP_ROW_FORM_DIRECT    1   /* <= 65536 rows: straight from x / i, once per block */
#define RSP_ROW_FORM_PARTITION 2   /* regrouped by (coarse) row block in one pass       */
#define RSP_ROW_FORM_TWO_LEVEL 3   /* regrouped in two passes                            */
#define RSP_ROW_FORM_SEGMENTS  4   /* table of column pieces per row block, no copy      */
int rsp_csc_row_sums(rsp_csc_t handle, double *sums);     /* nrow doubles, host */
int rsp_csc_row_means(rsp_csc_t handle, double *means);
int rsp_csc_row_form(rsp_csc_t handle);
size_t rsp_row_sums_workspace_bytes(int32_t nrow, int64_t nnz);
int rsp_row_sums_device(const double *d_x, const int32_t *d_i, int32_t nrow,
                        int64_t nnz, double *d_sums, void *d_workspace,
                        size_t workspace_bytes, void *stream);
int rsp_row_means_device(const double *d_x, const int32_t *d_i, int32_t nrow,
                         int32_t ncol, int64_t nnz, double *d_means,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- Matrix::crossprod ---------------------------------------------------- */
/*
 * Reference RcppSparse.h:159-194: the dense ncol x ncol matrix t(A) %*% A (column-
 * major, ncol * ncol doubles), each entry the sparse dot product of two columns over
 * their common rows in ascending row order.  Meant for matrices with few columns (the
 * output is O(ncol^2)).  Same order and separate multiply / add as the reference
 * loop: bit-identical results for finite data.  Needs the row indices (which must lie
 * in [0, nrow), as in any valid dgCMatrix).
 *
 * rsp_crossprod_device: with a workspace of rsp_crossprod_workspace_bytes(nrow, ncol,
 * nnz) bytes the row-major form of A is built in it and each result column is accumulated
 * by walking the rows its column touches (work = sum over rows of nnz(row)^2 products);
 * with d_workspace == NULL a scratch-free 64 x 64 tile kernel is used instead (slower on
 * sparse data).  Both give the same bits.  rsp_crossprod_workspace_bytes needs a usable
 * device; 0 = error.
 *
 * One exception to "same bits": for ncol <= 512 and columns of >= 4096 stored entries on
 * average (the tall matrices crossprod is meant for), the workspace form sums in a different
 * order -- rows are densified 64 (from 97 columns on: 32, from 257 on: 16) at a time and t(P) P runs on the matrix cores, every workgroup
 * over its own range of rows, results added in workgroup order: deterministic, within
 * 1e-12 * sum|x1 x2| per entry of the reference's order, and two to three orders of magnitude
 * faster than walking 48 columns of 4.5e7 rows one product after the other (7 ms against 24.8 s).  If x holds
 * a NaN or an infinity -- from 97 columns on: if any sum is not finite -- the bit-identical kernel does the work
 * instead (a structural zero must not meet a non-finite value).  rsp_set_crossprod_exact(1), or RSP_CROSSPROD_EXACT=1 in the
 * environment, keeps the bit-identical forms everywhere; set it before asking for the workspace size.
 * The matrix-core form is taken where a cost model says it pays (its time is that of a DENSE rank update, nrow x
 * ncol^2, however sparse the matrix: 256 columns of 4096 entries over 1e6 rows stay with the exact form, the same
 * columns over 1e5 rows or with 20000 entries each do not); rsp_crossprod_form tells which form sizes select.
 */
#define RSP_CROSSPROD_FORM_EXACT 0   /* reference order, bit-identical */
#define RSP_CROSSPROD_FORM_TALL  1   /* matrix cores, within 1e-12 * sum|x1 x2| */
int rsp_crossprod_form(int32_t nrow, int32_t ncol, int64_t nnz);   /* with a workspace; -1 = sizes out of range / no device */
int rsp_set_crossprod_exact(int exact);
int rsp_csc_crossprod(rsp_csc_t handle, double *out);        /* host, ncol*ncol */
size_t rsp_crossprod_workspace_bytes(int32_t nrow, int32_t ncol, int64_t nnz);
int rsp_crossprod_device(const double *d_x, const int32_t *d_i, const int32_t *d_p,
                         int32_t nrow, int32_t ncol, int64_t nnz, double *d_out,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- column-range partitioner (multi-GPU; pure integer, host) ---------- */
/*
 * nnz-balanced contiguous column ranges: bounds[k] = first column c with
 * p[c] >= k*nnz/nparts (k = 1..nparts-1), bounds[0] = 0, bounds[nparts] = ncol.
 * No column is split.  Part k owns columns [bounds[k], bounds[k+1]) and the
 * x range [p[bounds[k]], p[bounds[k+1]]).  p is host memory.
 */
int rsp_partition_columns(const int32_t *p, int32_t ncol, int32_t nparts,
                          int32_t *bounds);
/* p_local[j] = p[c0 + j] - p[c0] for j in [0, c1-c0] (rebased shard offsets). */
int rsp_rebase_offsets(const int32_t *p, int32_t c0, int32_t c1, int32_t *p_local);

/* ---- RCCL gatherv of per-shard sums over xGMI -------------------------- */
/* One communicator per process (one process per GPU).  The 128-byte unique id
 * is created on rank 0 and distributed by the caller (e.g. a torch.distributed
 * broadcast or a file), then every rank calls rsp_comm_init, which makes `device`
 * the calling thread's current HIP device (as a process-per-GPU program wants). */
#define RSP_UNIQUE_ID_BYTES 128
typedef struct rsp_comm *rsp_comm_t;

int rsp_comm_unique_id(void *id_bytes);
/* Which RCCL this process runs: *version = ncclGetVersion (e.g. 22606), library_path = the file "librccl*" is mapped
 * from (/proc/self/maps; "" if not found).  Either output may be NULL. */
int rsp_rccl_info(int *version, char *library_path, size_t capacity);
int rsp_comm_init(const void *id_bytes, int nranks, int rank, int device,
                  rsp_comm_t *comm);
/*
 * Gather counts[r] doubles from every rank r into d_recv + displs[r] on `root`
 * (grouped ncclSend / ncclRecv; RCCL has no native gatherv).  d_recv, counts
 * and displs are only read on root (counts/displs are host arrays of nranks
 * entries); other ranks pass their own send_count.  Enqueued on `stream`.
 */
int rsp_comm_gatherv(rsp_comm_t comm, const double *d_send, int64_t send_count,
                     double *d_recv, const int64_t *counts, const int64_t *displs,
                     int root, void *stream);
/*
 * Matrix::rowSums / rowMeans (reference RcppSparse.h:138-144, :151-156) over column-range shards: every rank
 * has computed the partial row sums of ITS columns (rsp_row_sums_device on its x / i slices: nrow doubles in
 * d_partial); the full sums are the partials added in rank order = column order, whatever the topology --
 * bit-identical to one process adding the shards' vectors in order (rsp_add_partials_device) and independent
 * of how RCCL would schedule a reduce.  Rows are cut into nranks slices, every rank receives its slice of
 * every other rank's vector (grouped ncclSend / ncclRecv, one pair per xGMI link), adds the nranks pieces in
 * rank order with one kernel, and the reduced slices are gathered to d_result on `root` (nrow doubles; only
 * read there).  ncol_for_means > 0 divides by it (rowMeans), 0 = sums.
 * Workspace: rsp_comm_reduce_rows_workspace_bytes(nranks, nrow), about (1 + 1 / nranks) * nrow * 8 bytes.
 * Enqueued on `stream`.
 */
size_t rsp_comm_reduce_rows_workspace_bytes(int nranks, int32_t nrow);
int rsp_comm_reduce_rows(rsp_comm_t comm, const double *d_partial, int32_t nrow,
                         int32_t ncol_for_means, double *d_result, void *d_workspace,
                         size_t workspace_bytes, int root, void *stream);
/* The add step on its own (one process holding several shards' partial vectors): d_out[j] =
 * ((part_0[j] + part_1[j]) + ...) + part_{nparts-1}[j], part_k = d_parts + k * stride, j in [0, n);
 * ncol_for_means as above. */
int rsp_add_partials_device(const double *d_parts, int32_t nparts, int64_t stride, int64_t n,
                            int32_t ncol_for_means, double *d_out, void *stream);
int rsp_comm_destroy(rsp_comm_t comm);

/* ---- direct-write gather (one process per GPU, one node): a comparator beside the RCCL gatherv ---------- */
/*
 * The root allocates its result buffer with rsp_shared_result_alloc and hands the 64-byte handle to the other rank
 * processes (any channel: a torch.distributed broadcast, a file); they map it with rsp_shared_result_open and
 * pass `mapped + displs[rank]` as d_sums to rsp_column_sums_device: their results are stored straight into the
 * root's memory, over xGMI, as the kernels produce them -- no send / receive pair, no staging.  What remains of the
 * exchange is one fence per call: every rank waits for its own stream, then the ranks cross a barrier
 * (rsp_host_barrier_*: a page of POSIX shared memory, a microsecond or two); after it the root owns a complete
 * vector.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment on hosts that offer dmabuf IPC only.
 * rsp_shared_result_close: owner != 0 frees the root's allocation, 0 unmaps a rank's view.
 */
#define RSP_IPC_HANDLE_BYTES 64
int rsp_shared_result_alloc(size_t bytes, void **d_ptr, void *handle_bytes);
int rsp_shared_result_open(const void *handle_bytes, void **d_ptr);
/* the gathered vector (or a part of it) into host memory -- the one D2H copy into the NumericVector; waits for `stream` */
int rsp_shared_result_read(const void *d_ptr, size_t offset_bytes, void *host, size_t bytes, void *stream);
int rsp_shared_result_close(void *d_ptr, int owner);
/*
 * The other comparator ("per-GPU D2H into disjoint slices of one pinned buffer"): ONE vector in POSIX
 * shared memory, mapped by every rank process and page-locked there (hipHostRegister), so that each rank's copy engine
 * writes the rank's slice straight into memory the root process reads -- over the rank's own host link, no xGMI hop, no
 * RCCL.  Per call: the rank's kernels, rsp_copy_to_host_async of its slice behind them (a copy kernel of the library where
 * the device can address the destination, the runtime's copy command otherwise), a wait for the own stream, one
 * host barrier; the root then holds the whole vector in HOST memory.  name: "/something", the same on every rank; the
 * creating rank passes create != 0, the others wait up to 30 s for it.  rsp_shared_host_close(ptr, bytes, name) also
 * unlinks the name (the creator), NULL just unmaps.
 */
int rsp_shared_host_open(const char *name, size_t bytes, int create, void **host_ptr);
int rsp_shared_host_close(void *host_ptr, size_t bytes, const char *unlink_name);
int rsp_copy_to_host_async(const double *d_src, double *host_dst, int64_t n, void *stream);
/* hipDeviceCanAccessPeer (1 for device == peer): bench.py asks before it lets kernels store across devices. */
int rsp_device_can_access_peer(int device, int peer, int *can);
typedef struct rsp_host_barrier *rsp_host_barrier_t;
/* name: a POSIX shared-memory name ("/something", the same on every rank); rank 0 creates it, the others wait for it */
int rsp_host_barrier_create(const char *name, int nranks, int rank, rsp_host_barrier_t *barrier);
int rsp_host_barrier_wait(rsp_host_barrier_t barrier, double timeout_seconds);
int rsp_host_barrier_destroy(rsp_host_barrier_t barrier);

/* ---- synthetic inputs (bench / tests) ---------------------------------- */
/*
 * d_x[k] = value(seed, first_idx + k) for k in [0, n): a counter-based
 * generator using integer arithmetic only, so any slice can be regenerated
 * bit-identically on the host.  kind 0: signed two-decimal values in
 * [-5.10, 5.10] ("rsparsematrix-like"); kind 1: U(0,1), all positive.
 */
int rsp_gen_values_device(double *d_x, int64_t n, uint64_t seed,
                          uint64_t first_idx, int kind, void *stream);

/* The k entries of column c get one row per stratum [r*nrow/k, (r+1)*nrow/k) (integer
 * arithmetic, position inside the stratum from the same hash): ascending, distinct rows --
 * valid dgCMatrix row indices for synthetic matrices, reproducible on the host. */
int rsp_gen_row_indices_device(int32_t *d_i, const int32_t *d_p, int32_t nrow,
                               int32_t ncol, uint64_t seed, void *stream);

/* ---- measurement and test knobs ------------------------------------------ */
/*
 * ONE entry for every process-wide knob (round 5; earlier rounds exported a setter each).  None of them is needed to
 * use the library: defaults are chosen per problem size, and results stay within the documented tolerance for every
 * value.  tools/ and the tests use them for A/B runs and to put a form on both sides of its threshold.  The same
 * names in upper case with the prefix RSP_ are read from the environment once, at first use (RSP_LEAN, RSP_TAPER =
 * "permille,rows", ...); a value set here wins.  Plans and handles sample the knobs when they are MADE; the
 * device-pointer entries read them at every call.  Keys:
 *   "chunk_rows"      128-element rows of x owned by one wavefront (0 = automatic; clamped to 1 GiB of x per chunk)
 *   "taper_permille"  the last value / 1000 of x is cut into shorter chunks, dispatched last (-1 default, 0 none) ...
 *   "taper_rows"      ... of this many rows (<= 0: default)
 *   "experiment"      alternative builds of the main kernel: 0 production, 1 = 16 rows in flight, 2 / 3 = 1 / 2
 *                     wavefronts per workgroup, 4 = default cache policy instead of nt loads, 5 = 4 rows in flight
 *   "lean"            plans: 0 never the lean form, 1 where it is the faster one (default), 2 wherever it applies
 *   "columns_form"    plans: 0 never, 1 where faster (default), 2 on every matrix whose longest column the kernel takes
 *   "row_segments"    handles' row sums: 0 never the segments form, 1 where faster (default), 2 wherever possible
 *   "row_slices"      row-restricted sums: 0 never the slice-major form, 1 where faster (default), 2 wherever possible
 *   "auto_plan"       rsp_column_sums_device / rsp_column_means_device plan for themselves (1, default) or never (0)
 *   "auto_min_nnz"    ... for matrices of at least this many entries (default 2^20; tests lower it to 1)
 *   "fold_fixup"      plain calls that are one round of waves: 1 = the fix-up runs inside the main launch (the last workgroup to
 *                     finish does it), 0 (default) = as a second launch.  Same bits; the folded form measured SLOWER (BASELINE
 *                     config 2: 37.0 against 22.5 us, profiles/DEAD_ENDS.md) and exists for that A/B
 *   read-only (rsp_debug_get): "auto_plans_made" / "auto_plans_freed" -- plans those entries have made / given up since the
 *                     process started; "auto_plans_retired" -- retired images still waiting for their events;
 *                     "auto_plans_recycled" -- plans that took a recycled allocation instead of a new one
 * An unknown key is RSP_ERR_BAD_ARG.  rsp_debug_get reads the value in force (environment and defaults resolved).
 * Threads: the knobs are atomics -- setting one while another thread is inside a call is safe and takes effect for
 * calls (plans, handles) that start afterwards; two libraries' worth of callers in one process share them, which is
 * why nothing a production caller needs goes through here.  The per-use switches are rsp_set_crossprod_exact (below
 * Matrix::crossprod) and rsp_csc_set_planned (per handle).
 * THREAD SAFETY OF THE LIBRARY: every entry may be called from any thread.  Calls on DIFFERENT handles, plans,
 * communicators or streams may run concurrently; calls on ONE handle / plan / communicator must not overlap (a
 * handle is owned by one thread at a time).  The one-shot host entries serialise per device (rsp_column_sums_host
 * holds its device's buffers for the call); the plan-free device entries take a process-wide mutex for the few
 * hundred nanoseconds of their bookkeeping.  rsp_last_error is per thread.
 */
int rsp_debug_set(const char *key, int value);
int rsp_debug_get(const char *key, int *value);
/* How a call over nnz entries is cut into chunks (one wavefront each) under the settings in force:
 * plan4 = { elements per body chunk, number of body chunks, elements per tail chunk, total chunks }.
 * Chunk w starts at w * body for w < nbody and at nbody * body + (w - nbody) * tail after that; all
 * sizes are multiples of 128 elements.  Pure host arithmetic, no device needed (tools, tests). */
int rsp_plan_describe(int64_t nnz, int32_t *plan4);

#ifdef __cplusplus
}
#endif
#endif /* RCPPSPARSE_HIP_H */

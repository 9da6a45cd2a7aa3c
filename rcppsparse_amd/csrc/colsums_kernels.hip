// colsums_kernels.hip -- CDNA4 (gfx950) segmented column-sum kernels.
//
// Replaces the double loop of reference src/example.cpp:28-30 (one
// Matrix::InnerIterator per column, reference inst/include/RcppSparse.h:218-233):
//
//     sums[c] = sum_{j = p[c]}^{p[c+1]-1} x[j]
//
// Design (MI355X-first):
//   * The unit of work is a *chunk* of x[] (a fixed number of 128-element rows),
//     not a set of columns, so skewed column lengths cannot unbalance the chip.
//     One wavefront owns one chunk and streams it with 16-byte-per-lane `nt`
//     buffer loads (1 KiB per wave instruction), BATCH_ROWS of them always in
//     flight in a rolling register pipeline.  i[] is never read by the sums.
//   * Column offsets p[] are staged per wave in an LDS window.
//   * A row with no column end inside takes the fast path: two v_add_f64 per lane.
//   * A row with 1-3 column ends: one masked DPP wave reduction per end.
//   * A group of 4 rows with many column ends (short and medium columns): staged in LDS and
//     handed out by column, 1 / 2 / 4 / 8 lanes per column; a lane adds its column's elements
//     in storage order (whole quads, then the last 0-3).
//   * Anything else (e.g. runs of empty columns): per-row LDS histogram of the ends ->
//     element ranks -> segmented DPP scan.
//   * Columns that cross chunk edges leave a head / tail partial and a 16-byte record per chunk;
//     a small second kernel adds those in ascending chunk order.  No floating-point atomics
//     anywhere: results are bit-stable run to run.
//   * Chunks: 256 rows for long calls with the last tenth of x in 64-row chunks (ChunkMap), 20
//     rows and 4 loads in flight for calls that fit one round of wavefronts (capi.hip make_plan).
//   * No MFMA: 1 FP64 add per 8 bytes, the bound is HBM bandwidth.
//   * The same template serves the "next" reductions: a per-element transform (sum of
//     squares / abs), max / min, or a row-set mask (streams i[] too and probes a row bitmap
//     in L1, in LDS or in L2 depending on its size).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "colsums_kernels.h"

namespace rsp {

// Diagnostic build only (`make stamps` -> librcppsparse_hip_stamps.so, never shipped): lane 0 of
// the first kStampChunks chunks records the constant 100 MHz clock at a few points of the main kernel
// into a buffer no other code reads (tools/stamps_report.py).
#ifdef RSP_STAMPS
__device__ unsigned long long g_stamps[kStampChunks * 8];
#define RSP_STAMP(k)                                                                        \
    do {                                                                                    \
        if (lane == 0 && w < kStampChunks) g_stamps[w * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
hipError_t read_stamps(unsigned long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long), 0,
                               hipMemcpyDeviceToHost);
}
#else
#define RSP_STAMP(k) do { } while (0)
#endif

typedef double d2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// lane-exchange helpers (wave64, DPP only: no LDS round trips)
// ---------------------------------------------------------------------------
// DPP controls used (gfx9 family encodings):
//   0xB1 quad_perm[1,0,3,2]   0x4E quad_perm[2,3,0,1]   0x141 row_half_mirror   0x140 row_mirror
//   0x111/0x112/0x114/0x118 row_shr:1/2/4/8 (inside each row of 16 lanes)
//   0x142 row_bcast:15 (lane 15 of a row -> every lane of the next row; row_mask 0xA = rows 1,3)
//   0x143 row_bcast:31 (lane 31 -> rows 2,3; row_mask 0xC)
//   0x138 wave_shr:1, 0x130 wave_shl:1 (whole-wave shift by one lane)
// Workgroups are dealt to the 8 XCDs round-robin (b mod 8), so neighbouring chunks sit in different L2s, and a
// 128-byte line of results that two chunks share is written back twice, as two partial lines.  Workgroup b of n ->
// the position it takes when every XCD is to work on RUNS of G neighbouring positions (the runs themselves still go
// round the XCDs, which keeps the reads spread over the address space).  Used by the lean planned kernel, whose chunks
// write only ~3 lines each (kLeanXcdRun); the last, partial round of runs keeps its order.
template <int G>
__device__ __forceinline__ int xcd_runs(int b, int n) {
    const int nfull = n / (8 * G) * (8 * G);
    if (b >= nfull) return b;
    const int k = b & 7, j = b >> 3;
    return ((j / G) * 8 + k) * G + j % G;
}

template <int CTRL, int ROWMASK = 0xF, bool ZERO_FILL = true>
__device__ __forceinline__ int dpp_i32(int v, int old = 0) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xF, ZERO_FILL);
}

template <int CTRL, int ROWMASK = 0xF, bool ZERO_FILL = true>
__device__ __forceinline__ double dpp_f64(double v) {
    // one DPP move per 32-bit half; lanes with no source (or a masked row) read +0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, ZERO_FILL);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, ZERO_FILL);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// How a column's terms are combined.  Everything on the hot path is a sum (identity +0.0);
// the generic column reduction also offers max / min of the stored
// entries (identity -inf / +inf, NaN entries are skipped like a `if (v > acc) acc = v` loop).
template <bool MEANS_, int OP_>
struct Policy {
    static constexpr bool kMeans = MEANS_;
    static constexpr int kOp = OP_;
    static constexpr bool kMax = (OP_ == kOpMax), kMin = (OP_ == kOpMin);
    static constexpr bool kSum = !kMax && !kMin;
    __device__ static __forceinline__ double id() {
        return kSum ? 0.0 : (kMax ? -__builtin_huge_val() : __builtin_huge_val());
    }
    __device__ static __forceinline__ double comb(double a, double b) {
        return kSum ? a + b : (kMax ? __builtin_fmax(a, b) : __builtin_fmin(a, b));
    }
    // value stored for a finished column
    __device__ static __forceinline__ double finish(double v, double divisor) {
        if (kSum) v += 0.0;   // a sum of -0.0 terms comes out +0.0 like the reference's 0.0-initialised accumulator
        if (kMeans) v = v / divisor;   // RcppSparse.h:147-148  sums[i] / Dim[0]
        return v;
    }
};

// Combination over all 64 lanes, returned wave-uniform.  Fixed tree: bit-stable.
template <class P>
__device__ __forceinline__ double wave_allreduce(double v) {
    v = P::comb(v, dpp_f64<0xB1>(v));    // xor 1            (every lane has a source in these four)
    v = P::comb(v, dpp_f64<0x4E>(v));    // xor 2
    v = P::comb(v, dpp_f64<0x141>(v));   // the two quads of each 8
    v = P::comb(v, dpp_f64<0x140>(v));   // the two halves of each row of 16: every lane holds its row's result
    return P::comb(P::comb(P::comb(readlane_f64(v, 0), readlane_f64(v, 16)), readlane_f64(v, 32)),
                   readlane_f64(v, 48));
}

// Inclusive prefix sum over the 64 lanes (int): 4 in-row steps + 2 row broadcasts.
__device__ __forceinline__ int wave_inclusive_scan_i32(int v) {
    v += dpp_i32<0x111>(v);
    v += dpp_i32<0x112>(v);
    v += dpp_i32<0x114>(v);
    v += dpp_i32<0x118>(v);
    v += dpp_i32<0x142, 0xA, false>(v);
    v += dpp_i32<0x143, 0xC, false>(v);
    return v;
}

// One step of the segmented inclusive scan: add the source lane's running sum when it
// carries the same key.  Keys are non-negative and non-decreasing across lanes; a lane
// with no source sees key -1 (never equal).
template <class P, int CTRL, int ROWMASK, bool ZERO_FILL>
__device__ __forceinline__ void seg_scan_step(double& X, int key) {
    const double Xs = dpp_f64<CTRL, ROWMASK, ZERO_FILL>(X);   // (fill value never used: key -1 below)
    const int ks = __builtin_amdgcn_update_dpp(-1, key, CTRL, ROWMASK, 0xF, false);
    if (ks == key) X = P::comb(X, Xs);
}

template <class P>
__device__ __forceinline__ double wave_segmented_inclusive_scan(double X, int key) {
    seg_scan_step<P, 0x111, 0xF, true>(X, key);
    seg_scan_step<P, 0x112, 0xF, true>(X, key);
    seg_scan_step<P, 0x114, 0xF, true>(X, key);
    seg_scan_step<P, 0x118, 0xF, true>(X, key);
    seg_scan_step<P, 0x142, 0xA, false>(X, key);   // previous row's last lane
    seg_scan_step<P, 0x143, 0xC, false>(X, key);   // lane 31 into rows 2 and 3
    return X;
}

// ---------------------------------------------------------------------------
// per-wave LDS window over p[]
// ---------------------------------------------------------------------------
// win[k] = p[min(wbase + k, ncol)] for k in [0, kPWin).  Consumers must check
// idx <= ncol themselves (entries past ncol repeat p[ncol]).
__device__ __forceinline__ void fill_window(int32_t* win, const int32_t* __restrict__ p,
                                            int wbase, int ncol, int lane) {
    int32_t t[kPWin / 64];
#pragma unroll
    for (int j = 0; j < kPWin / 64; ++j) {
        uint32_t idx = (uint32_t)wbase + (uint32_t)(j * 64 + lane);
        if (idx > (uint32_t)ncol) idx = (uint32_t)ncol;
        t[j] = p[idx];
    }
    __builtin_amdgcn_wave_barrier();   // earlier reads of the old window stay above
#pragma unroll
    for (int j = 0; j < kPWin / 64; ++j) win[j * 64 + lane] = t[j];
    __builtin_amdgcn_wave_barrier();
}

struct WaveState {
    int ccur;            // column that owns the current stream position
    int qnext;           // p[ccur + 1]  (end of that column); valid iff has_next
    int wbase;           // p-index of win[0]
    bool has_next;       // ccur + 1 <= ncol
    bool head_open;      // no column end seen in this chunk yet
    bool head_complete;  // the chunk starts exactly at p[c0]
    bool acc_in_lane0;   // the running result of the open column sits in lane 0's acc0 alone
                         // (every other lane's acc0 and all of acc1 hold the identity)
};

// Make sure win covers p-indices [k, k + need).
__device__ __forceinline__ void ensure_window(WaveState& st, int32_t* win,
                                              const int32_t* __restrict__ p, int k, int need,
                                              int ncol, int lane) {
    if (k < st.wbase || k + need > st.wbase + kPWin) {
        st.wbase = k;
        fill_window(win, p, k, ncol, lane);
    }
}

// Per-element transform of the generic column reduction: the same
// InnerIterator-shaped loop with a different body -- sum, sum of squares, sum of |x|.
// f(0) = 0 for the sums, so the zero-filled out-of-range lanes stay harmless (max / min mask them).
template <int OP>
__device__ __forceinline__ double xf(double v) {
    if (OP == kOpSumSquares) return v * v;
    if (OP == kOpSumAbs) return __builtin_fabs(v);
    // max / min skip NaN entries (the loop `if (v > acc) acc = v` never takes one): turn them
    // into the identity up front so that every path, also a one-element column, agrees
    if (OP == kOpMax) return (v != v) ? -__builtin_huge_val() : v;
    if (OP == kOpMin) return (v != v) ? __builtin_huge_val() : v;
    return v;
}

// ---------------------------------------------------------------------------
// slow path: one 128-element row that contains >= 1 column end
// ---------------------------------------------------------------------------
// Lane l holds elements e0 = rs + 2l (v0) and e1 = e0 + 1 (v1).  On entry
// acc0/acc1 are the per-lane partial sums of column st.ccur from earlier rows.
template <class P>
__device__ __forceinline__ void slow_row(double v0, double v1, int rs, int lane, WaveState& st,
                                      double& acc0, double& acc1, int32_t* win, int32_t* hist,
                                      const int32_t* __restrict__ p, int ncol, int w,
                                      double* __restrict__ out, double* __restrict__ carry_head,
                                      double divisor) {
    // 1. fold the running per-lane partials into the first element of the row
    const double A = wave_allreduce<P>(P::comb(acc0, acc1));
    if (lane == 0) v0 = P::comb(v0, A);

    // 2. histogram of column ends q in (rs, rs + 128]:  hist[q - rs]++
    *(int2*)&hist[2 * lane] = make_int2(0, 0);
    if (lane == 0) hist[128] = 0;
    __builtin_amdgcn_wave_barrier();
    int k = st.ccur + 1;
    for (;;) {
        ensure_window(st, win, p, k, 64, ncol, lane);
        const uint32_t idx = (uint32_t)k + (uint32_t)lane;
        const bool valid = idx <= (uint32_t)ncol;
        const int q = win[(valid ? (int)idx : k) - st.wbase];
        const uint32_t d = (uint32_t)q - (uint32_t)rs;
        const bool inrow = valid && (d - 1u) < 128u;
        if (inrow) atomicAdd(&hist[d], 1);
        const int n = __popcll(__ballot(inrow));
        k += n;
        if (n < 64) break;
    }
    const int tot = k - (st.ccur + 1);   // column ends in this row
    __builtin_amdgcn_wave_barrier();

    // 3. rank of each element = number of ends at or before it
    const int2 h = *(const int2*)&hist[2 * lane];
    const int S = wave_inclusive_scan_i32(h.x + h.y);
    const int kR = S;          // rank of e1
    const int kL = S - h.y;    // rank of e0

    // 4. segmented inclusive scan (key kR) of each lane's open-right part
    const bool split = kL != kR;              // a column ends between e0 and e1
    const double X = wave_segmented_inclusive_scan<P>(split ? v1 : P::comb(v0, v1), kR);

    // 5. finished segments
    const double Xp = dpp_f64<0x138>(X);                               // lane - 1
    const int kp = __builtin_amdgcn_update_dpp(-1, kR, 0x138, 0xF, 0xF, false);
    const double totalL = P::comb(v0, (kp == kL) ? Xp : P::id());      // segment ending at e0
    int kN = dpp_i32<0x130>(kL);                                       // lane + 1
    if (lane == 63) kN = tot;
    const bool endR = kN > kR;                                         // segment ending at e1
    const int cbase = st.ccur;

    if (split) {
        const int c = cbase + kL;
        if (st.head_open && kL == 0) {
            carry_head[w] = totalL;
            if (st.head_complete && c < ncol) out[c] = P::finish(totalL, divisor);
        } else if (c < ncol) {
            out[c] = P::finish(totalL, divisor);
        }
    }
    if (endR) {
        const int c = cbase + kR;
        if (st.head_open && kR == 0) {
            carry_head[w] = X;
            if (st.head_complete && c < ncol) out[c] = P::finish(X, divisor);
        } else if (c < ncol) {
            out[c] = P::finish(X, divisor);
        }
    }

    // empty columns (rank jumps by more than one): zero-fill, whole wave per gap
    const int gapL = split ? (kR - kL - 1) : 0;
    const int gapR = endR ? (kN - kR - 1) : 0;
    uint64_t mL = __ballot(gapL > 0);
    uint64_t mR = __ballot(gapR > 0);
    while (mL) {
        const int l = __builtin_ctzll(mL);
        mL &= mL - 1;
        const int start = __builtin_amdgcn_readlane(cbase + kL + 1, l);
        const int cnt = __builtin_amdgcn_readlane(gapL, l);
        for (int c = lane; c < cnt; c += 64)
            if (start + c < ncol) out[start + c] = P::finish(P::id(), divisor);
    }
    while (mR) {
        const int l = __builtin_ctzll(mR);
        mR &= mR - 1;
        const int start = __builtin_amdgcn_readlane(cbase + kR + 1, l);
        const int cnt = __builtin_amdgcn_readlane(gapR, l);
        for (int c = lane; c < cnt; c += 64)
            if (start + c < ncol) out[start + c] = P::finish(P::id(), divisor);
    }

    // 6. carry the open tail of the row and advance the column cursor
    acc0 = (lane == 63 && !endR) ? X : P::id();
    acc1 = P::id();
    st.acc_in_lane0 = false;
    st.ccur = cbase + tot;
    st.head_open = false;
    st.has_next = (uint32_t)k <= (uint32_t)ncol;
    if (st.has_next) {
        ensure_window(st, win, p, k, 1, ncol, lane);
        st.qnext = __builtin_amdgcn_readfirstlane(win[k - st.wbase]);
    }
}

// ---------------------------------------------------------------------------
// rows with only a few column ends, and groups of rows with many
// ---------------------------------------------------------------------------
// Per-lane view of the next 64 column ends: lane j holds q = p[k + j] (valid iff
// k + j <= ncol), read from the LDS window.
__device__ __forceinline__ int load_next_ends(WaveState& st, int32_t* win,
                                              const int32_t* __restrict__ p, int k, int ncol, int lane,
                                              bool& valid) {
    // keep p[k - 1] (the start of the column whose end is p[k]) in the window as well, so the
    // dense path, which needs the column starts, does not have to refill right after this
    ensure_window(st, win, p, k - 1, 65, ncol, lane);
    const uint32_t idx = (uint32_t)k + (uint32_t)lane;
    valid = idx <= (uint32_t)ncol;
    return win[(valid ? (int)idx : k) - st.wbase];
}

// After `k - 1` became the current column: refresh qnext / has_next.
__device__ __forceinline__ void refresh_next(WaveState& st, int32_t* win, const int32_t* __restrict__ p,
                                             int k, int ncol, int lane) {
    st.has_next = (uint32_t)k <= (uint32_t)ncol;
    if (st.has_next) {
        ensure_window(st, win, p, k, 1, ncol, lane);
        st.qnext = __builtin_amdgcn_readfirstlane(win[k - st.wbase]);
    }
}

template <class P>
__device__ __forceinline__ void emit_column(const WaveState& st, int c, int rel, double total, int ncol,
                                            int w, double* __restrict__ out,
                                            double* __restrict__ carry_head, double divisor) {
    // rel = index of this column end counted from st.ccur; the chunk's first end is the head
    if (st.head_open && rel == 0) {
        carry_head[w] = total;
        if (st.head_complete && c < ncol) out[c] = P::finish(total, divisor);
    } else if (c < ncol) {
        out[c] = P::finish(total, divisor);
    }
}

// Row with n in [1, kFewEnds] column ends: one masked wave reduction per end (about 40
// instructions each) instead of the general rank + segmented-scan machinery.
template <class P>
__device__ __forceinline__ void few_ends_row(double v0, double v1, int rs, int lane, int n, int wq,
                                             WaveState& st, double& acc0, double& acc1, int32_t* win,
                                             const int32_t* __restrict__ p, int ncol, int w,
                                             double* __restrict__ out, double* __restrict__ carry_head,
                                             double divisor) {
    const int o0 = 2 * lane;   // row-relative offsets of this lane's two elements
    int lo = 0;                // row-relative start of the segment being closed
    double a = P::comb(acc0, acc1);
    for (int j = 0; j < n; ++j) {
        const int d = __builtin_amdgcn_readlane(wq, j) - rs;   // end offset in (0, 128], uniform
        const double t0 = (o0 >= lo && o0 < d) ? v0 : P::id();
        const double t1 = (o0 + 1 >= lo && o0 + 1 < d) ? v1 : P::id();
        const double total = wave_allreduce<P>(P::comb(a, P::comb(t0, t1)));
        if (lane == 0) emit_column<P>(st, st.ccur + j, j, total, ncol, w, out, carry_head, divisor);
        a = P::id();
        lo = d;
    }
    acc0 = (o0 >= lo) ? v0 : P::id();
    acc1 = (o0 + 1 >= lo) ? v1 : P::id();
    st.acc_in_lane0 = false;
    st.ccur += n;
    st.head_open = false;
    refresh_next(st, win, p, st.ccur + 1, ncol, lane);
}

// Dense group: kGroupRows consecutive rows (512 elements) holding several column ends.  The
// rows are staged in LDS and the work is handed out by *column*, L lanes per column with
// L = 1 / 2 / 4 / 8 for an average segment of up to 16 / 32 / 64 / more elements:
//   L = 1 (short columns): lane j takes column ccur + j, reads its bounds from the p window and
//     adds its elements from LDS sequentially -- exactly the reference's storage order, so a
//     column that lives inside one group comes out bit-identical to the reference loop, and the
//     open column at either end of the group continues the same sequential chain through the
//     carry.  More than 64 segments are taken 64 at a time.
//   L > 1 (medium columns): the L lanes of a column each add every L-th element in storage
//     order and the L partial results are combined by a fixed DPP tree (deterministic; no
//     longer the reference's order, like every other path for columns of that length).
// No bitmap, no scans; empty columns are just zero-length ranges.  The group goes to the row
// paths instead (returns false, nothing consumed) if a lane would have to add more than
// kDenseMaxLen elements alone (it would hold the wave up).
template <class P>
__device__ __forceinline__ bool dense_group(const d2 (&v)[kGroupRows], int gs, uint32_t glim, int n_ends,
                                            int lane, WaveState& st, double& acc0, double& acc1,
                                            int32_t* win, double* stage, const int32_t* __restrict__ p,
                                            int ncol, int w, double* __restrict__ out,
                                            double* __restrict__ carry_head, double divisor) {
    const int ge = gs + (int)glim;   // one past the last element of the group this chunk owns
    // segment j = 0 is the open column continuing into the group, j = n_ends the column still
    // open at the group's end: n_ends + 1 segments
    // Lanes per column from the average segment length glim / nseg (uniform): up to 16 elements
    // -> 1 lane (short columns stay in reference order, also in the partial group at the end of
    // a chunk), up to 32 -> 2, up to 64 -> 4, longer -> 8.  glim <= 512, so each bound also
    // limits the number of segments: nseg << shift <= 64 for shift > 0.
    const uint32_t nseg = (uint32_t)n_ends + 1u;
    const int shift = glim <= 16u * nseg ? 0 : (glim <= 32u * nseg ? 1 : (glim <= 64u * nseg ? 2 : 3));   // log2(L)
    const int L = 1 << shift;

    // stage the four rows: element e of the group at stage[e]
#pragma unroll
    for (int r = 0; r < kGroupRows; ++r)
        *(d2*)&stage[r * kRowElems + 2 * lane] = v[r];
    // result so far of the column open at gs: after a dense group it already sits in lane 0
    const double A = st.acc_in_lane0 ? readlane_f64(acc0, 0) : wave_allreduce<P>(P::comb(acc0, acc1));
    __builtin_amdgcn_wave_barrier();

    // A lane must not add more than kDenseMaxLen elements alone.  With 8 lanes per column it cannot
    // (512 / 8); with 64 or more ends the average segment is at most 8 elements and a rare long one
    // only costs its own length once.  In between the bounds of the (single) pass are checked
    // before anything is emitted.
    const bool check_len = shift < 3 && n_ends < 64;

    double carry_out = P::id();
    int owner;   // lane holding the running result of the column still open at the group's end
    if (shift == 0) {
        for (int j0 = 0; j0 <= n_ends; j0 += 64) {
            ensure_window(st, win, p, st.ccur + j0, 66, ncol, lane);
            const int woff = st.ccur - st.wbase;
            const int j = j0 + lane;
            const bool active = j <= n_ends;
            int lo = 0, hi = 0;
            if (active) {
                lo = ((j == 0) ? gs : win[woff + j]) - gs;
                hi = ((j == n_ends) ? ge : win[woff + j + 1]) - gs;
                // a valid p[] gives 0 <= lo <= hi <= glim; an invalid one (device entries do not check it)
                // must not turn into a long loop or a read outside the staged group
                lo = lo < 0 ? 0 : (lo > (int)glim ? (int)glim : lo);
                hi = hi < lo ? lo : (hi > (int)glim ? (int)glim : hi);
            }
            if (check_len && __ballot(hi - lo > kDenseMaxLen) != 0ull) return false;   // (single pass: nothing emitted yet)
            double s = (j == 0) ? A : P::id();   // the continuing column keeps adding to its running result
            // Storage-order adds.  Whole quads first: four unguarded LDS reads, four adds and ONE select
            // per lane and step (a lane whose column has no whole quad left keeps its sum; what it read
            // past its column's end stays inside the workgroup's LDS and is thrown away); then the last
            // 0-3 elements of every column.  Same order of adds as element by element, at less than half
            // the vector instructions.
            const int n = hi - lo;
            const double* sp = stage + lo;
            const int nquads = n >> 2;
            for (int q = 0; __ballot(q < nquads) != 0ull; ++q) {
                const double e0 = sp[4 * q], e1 = sp[4 * q + 1], e2 = sp[4 * q + 2], e3 = sp[4 * q + 3];
                const double t = P::comb(P::comb(P::comb(P::comb(s, e0), e1), e2), e3);
                s = q < nquads ? t : s;
            }
            {
                const double* tp = sp + 4 * nquads;
                const int rem = n & 3;
                const double e0 = tp[0], e1 = tp[1], e2 = tp[2];
                const double t0 = P::comb(s, e0), t1 = P::comb(t0, e1), t2 = P::comb(t1, e2);
                s = rem == 0 ? s : (rem == 1 ? t0 : (rem == 2 ? t1 : t2));
            }
            if (active && j < n_ends) emit_column<P>(st, st.ccur + j, j, s, ncol, w, out, carry_head, divisor);
            if (active && j == n_ends) carry_out = s;
        }
        owner = n_ends & 63;
    } else {
        // (n_ends + 1) * L <= 64: one pass, lanes [j * L, (j + 1) * L) share column ccur + j
        ensure_window(st, win, p, st.ccur, 66, ncol, lane);
        const int woff = st.ccur - st.wbase;
        const int j = lane >> shift, sub = lane & (L - 1);
        const bool active = j <= n_ends;
        int lo = 0, hi = 0;
        if (active) {
            lo = ((j == 0) ? gs : win[woff + j]) - gs;
            hi = ((j == n_ends) ? ge : win[woff + j + 1]) - gs;
            lo = lo < 0 ? 0 : (lo > (int)glim ? (int)glim : lo);   // (see above: only an invalid p[] is clamped)
            hi = hi < lo ? lo : (hi > (int)glim ? (int)glim : hi);
        }
        if (check_len && __ballot(hi - lo > (kDenseMaxLen << shift)) != 0ull) return false;
        double s = (lane == 0) ? A : P::id();
        const int cnt = (hi - lo - sub + L - 1) >> shift;   // this lane adds elements lo + sub + m * L, m < cnt
        const double* sp = stage + lo + sub;
        const int nquads = cnt >> 2;                        // whole quads first, then the last 0-3 (see above)
        for (int q = 0; __ballot(q < nquads) != 0ull; ++q) {
            const int k = 4 * q;
            const double e0 = sp[k << shift], e1 = sp[(k + 1) << shift], e2 = sp[(k + 2) << shift],
                         e3 = sp[(k + 3) << shift];
            const double t = P::comb(P::comb(P::comb(P::comb(s, e0), e1), e2), e3);
            s = q < nquads ? t : s;
        }
        {
            const int k = 4 * nquads, rem = cnt & 3;
            const double e0 = sp[k << shift], e1 = sp[(k + 1) << shift], e2 = sp[(k + 2) << shift];
            const double t0 = P::comb(s, e0), t1 = P::comb(t0, e1), t2 = P::comb(t1, e2);
            s = rem <= 0 ? s : (rem == 1 ? t0 : (rem == 2 ? t1 : t2));
        }
        // the L partial results of a column: pairs, quads, the two quads of each 8
        s = P::comb(s, dpp_f64<0xB1>(s));
        if (shift >= 2) s = P::comb(s, dpp_f64<0x4E>(s));
        if (shift >= 3) s = P::comb(s, dpp_f64<0x141>(s));
        if (active && sub == 0 && j < n_ends)
            emit_column<P>(st, st.ccur + j, j, s, ncol, w, out, carry_head, divisor);
        carry_out = s;
        owner = n_ends << shift;
    }
    // hand the open column's running sum to whatever comes next (one lane holds it)
    const double co = readlane_f64(carry_out, owner);
    acc0 = (lane == 0) ? co : P::id();
    acc1 = P::id();
    st.acc_in_lane0 = true;
    st.ccur += n_ends;
    if (n_ends > 0) st.head_open = false;
    refresh_next(st, win, p, st.ccur + 1, ncol, lane);
    return true;
}

// One row: fast path, few-ends path or the general slow path.
template <class P>
__device__ __forceinline__ void process_row(double v0, double v1, int rs, int lane, WaveState& st,
                                       double& acc0, double& acc1, int32_t* win, int32_t* hist,
                                       const int32_t* __restrict__ p, int ncol, int w,
                                       double* __restrict__ out, double* __restrict__ carry_head,
                                       double divisor) {
    const uint32_t dq = (uint32_t)st.qnext - (uint32_t)rs;
    if (st.has_next && (dq - 1u) < 128u) {
        bool valid;
        const int wq = load_next_ends(st, win, p, st.ccur + 1, ncol, lane, valid);
        const uint32_t d = (uint32_t)wq - (uint32_t)rs;
        const int n = __popcll(__ballot(valid && (d - 1u) < 128u));
        if (n <= kFewEnds)
            few_ends_row<P>(v0, v1, rs, lane, n, wq, st, acc0, acc1, win, p, ncol, w, out, carry_head,
                                divisor);
        else
            slow_row<P>(v0, v1, rs, lane, st, acc0, acc1, win, hist, p, ncol, w, out, carry_head,
                            divisor);
    } else {
        acc0 = P::comb(acc0, v0);
        acc1 = P::comb(acc1, v1);
        st.acc_in_lane0 = false;
    }
}

// ---------------------------------------------------------------------------
// main kernel: one wavefront per chunk
// ---------------------------------------------------------------------------
// LDSMAP (row-restricted sums only): the row bitmap is copied into (dynamic) LDS once per workgroup and
// probed there.  With ~1e6 rows the bitmap (125 KB) is too big for the 32 KB L1, and 64 random 4-byte
// probes per wave instruction into L2 cost more than the 12 B/nnz stream itself; in LDS they are a
// couple of cycles.  One workgroup (WPG wavefronts) per CU then.
// PLANNED (inspector-executor, rsp_column_sums_plan_*): a host-side inspector that has seen p[] hands every
// chunk a record {c0, xs0}: the first element it owns (xs0 = the first column start at or after the chunk's
// grid position, at most one group of 512 elements into it) and the column that starts there.  Chunk w then owns
// exactly the columns [c0_w, c0_{w+1}) = the elements [xs0_w, xs0_{w+1}): it gives the identity to what
// precedes xs0 (the previous chunk's), streams its grid rows exactly like the general kernel, and finishes the
// one column that reaches past its grid end with a single extra load (the `tail`, requested as soon as the
// records are there) and one wave reduction.  No column search (four dependent round trips before the first
// add in a short call), no carries, no fix-up launch.  The grid loads start immediately, before the records
// have arrived.
template <class P, bool COHERENT>
__device__ __forceinline__ void fixup_chunk(int w, int lane, int32_t ncol, int32_t nchunks, double* __restrict__ out,
                                            const double* carry_head, const double* carry_tail, const int4* carry_info,
                                            double divisor);

template <int BATCH_ROWS, bool MEANS, int AUX, int WPG = kWavesPerWG, int OP = kOpSum, bool LDSMAP = false,
          bool PLANNED = false, bool FOLD = false>
__global__ __launch_bounds__(WPG * 64) void colsums_chunks_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ p, int32_t ncol, int32_t nnz,
    ChunkMap cmap, int32_t nchunks, double* __restrict__ out,
    double* __restrict__ carry_head, double* __restrict__ carry_tail,
    int4* __restrict__ carry_info, double divisor, const int32_t* __restrict__ rows_i,
    const uint32_t* __restrict__ row_bitmap, int32_t bitmap_words, const int2* __restrict__ plan_rec = nullptr,
    const int32_t* __restrict__ run_if = nullptr, uint32_t* __restrict__ ticket = nullptr) {
    static_assert(BATCH_ROWS % kGroupRows == 0, "batch must be whole groups");
    static_assert(!(FOLD && PLANNED), "a planned launch has no carries to fix up");
    typedef Policy<MEANS, OP> P;
    constexpr bool MASKED = (OP == kOpMaskedIn || OP == kOpMaskedOut);
    // row-restricted sums: the slice-major form (colsums_rowslices.hip) ran instead unless its guard said otherwise
    if (MASKED && run_if != nullptr && *run_if == 0) return;
    __shared__ __attribute__((aligned(16))) double s_stage[WPG][kStageSlots];
    __shared__ __attribute__((aligned(16))) int32_t s_win[WPG][kPWin];
    static_assert(kHistPad * sizeof(int32_t) <= kStageSlots * sizeof(double), "the histogram lives in the staging area");

    extern __shared__ uint32_t s_bitmap[];   // LDSMAP only: bitmap_words words
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = blockIdx.x * WPG + wave_in_wg;
    if (LDSMAP) {   // (every wavefront of the workgroup takes part, also one without a chunk)
        for (int k = threadIdx.x; k < bitmap_words; k += WPG * 64) s_bitmap[k] = row_bitmap[k];
        __syncthreads();
    }
    // (FOLD: a wavefront without a chunk -- the last workgroup may have some -- still meets the others at the barriers below)
    if (!FOLD && w >= nchunks) return;
    if (w < nchunks) {
    int32_t* win = s_win[wave_in_wg];
    double* stage = s_stage[wave_in_wg];
    // the general row path's histogram shares the staging area of the dense path: a wave is in one
    // of the two at any time, and a dense group that gives up has no further use for what it staged.
    // 21.5 KB of LDS per workgroup = 7 workgroups (28 wavefronts) per CU.
    int32_t* hist = reinterpret_cast<int32_t*>(stage);

    RSP_STAMP(0);
    const int32_t cs = (int32_t)cmap.start(w);          // (< nnz <= 2^31 - 1)
    const int64_t ce64 = (int64_t)cs + cmap.elems(w);
    const int32_t ce = ce64 < (int64_t)nnz ? (int32_t)ce64 : nnz;
    const int32_t nrows = (int32_t)(((int64_t)ce - cs + 127) >> 7);
    int2 rec = make_int2(0, 0), rec_next = make_int2(0, 0);
    if (PLANNED) {
        rec = plan_rec[w];
        rec_next = plan_rec[w + 1];
    }

    const double* xb = x + cs;
    const uint32_t xbytes = (uint32_t)(ce - cs) * 8u;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)xbytes, 0x00020000);
    const int voff = lane * 16;

    // masked reductions also stream the row indices of the chunk (4 B/nnz) and probe a
    // row bitmap (L2-resident: nrow / 8 bytes) when a row of x is consumed
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(MASKED ? rows_i + cs : nullptr), 0, MASKED ? (int)(xbytes >> 1) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)row_bitmap, 0, MASKED ? bitmap_words * 4 : 0, 0x00020000);

    d2 v[BATCH_ROWS];
    int2 iv[MASKED ? BATCH_ROWS : 1];
#pragma unroll
    for (int r = 0; r < BATCH_ROWS; ++r) {
        v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, r * 1024, AUX));
        if (MASKED)
            iv[r] = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(ir, lane * 8, r * 512, AUX));
    }

    int lo = 0, hi = ncol;   // invariant: p[lo] <= cs < p[hi]
    if (PLANNED) {
        lo = __builtin_amdgcn_readfirstlane(rec.x);   // the inspector's answer: nothing to search
        hi = lo + 1;
    } else if (BATCH_ROWS == 4 && ncol >= kGuessWindow) {
        // Short calls (one round of waves, nothing to hide the search behind): guess the column from
        // "all columns equally long" and read kGuessWindow offsets around the guess in ONE round trip.
        // For uniform matrices the chunk's first column is in there (C2: within +-160 columns in 95 % of the
        // chunks); otherwise the window only narrows the range for the search below.
        int base = (int)(((int64_t)cs * ncol) / nnz) - kGuessWindow / 2;
        base = base < 0 ? 0 : (base > ncol - (kGuessWindow - 1) ? ncol - (kGuessWindow - 1) : base);
        int below = 0;   // window entries <= cs (a prefix of the window: p is non-decreasing)
        int32_t pv[kGuessWindow / 64];
#pragma unroll
        for (int k = 0; k < kGuessWindow / 64; ++k) pv[k] = p[base + k * 64 + lane];
#pragma unroll
        for (int k = 0; k < kGuessWindow / 64; ++k) below += __popcll(__ballot(pv[k] <= cs));
        if (below == 0) {
            hi = base > 0 ? base : 1;          // (p[0] = 0 <= cs for a valid matrix)
        } else if (below >= kGuessWindow) {
            lo = base + kGuessWindow - 1;
        } else {
            lo = base + below - 1;
            hi = lo + 1;
        }
    }
    while (hi - lo > 1) {
        const int step = (int)(((int64_t)hi - lo + 63) >> 6);
        const int64_t j = (int64_t)lo + (int64_t)(lane + 1) * step;
        const bool valid = j < hi;
        const int pv = p[valid ? j : hi];
        const bool le = valid && pv <= cs;
        const int n = __popcll(__ballot(le));
        const int64_t nlo = (int64_t)lo + (int64_t)n * step;
        const int64_t nhi = nlo + step;
        lo = (int)nlo;
        hi = nhi < hi ? (int)nhi : hi;
    }
    const int c0 = lo;
    const int32_t xs0 = PLANNED ? __builtin_amdgcn_readfirstlane(rec.y) : cs;   // first element this chunk owns
    // PLANNED: the column open at the grid end reaches `ext` elements (at most one group) past it; they are
    // requested now and added after the last grid row
    const int32_t c_end = PLANNED ? __builtin_amdgcn_readfirstlane(rec_next.x) : 0;   // first column of the next chunk
    const int32_t ext = PLANNED ? __builtin_amdgcn_readfirstlane(rec_next.y) - ce : 0;
    d2 tail[PLANNED ? kGroupRows : 1];
    if (PLANNED) {
        const __amdgpu_buffer_rsrc_t tr =
            __builtin_amdgcn_make_buffer_rsrc((void*)(x + ce), 0, ext > 0 ? ext * 8 : 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < kGroupRows; ++r)   // (rows past `ext` cost no traffic: the bounds check answers them)
            tail[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(tr, voff, r * 1024, AUX));
    }
    RSP_STAMP(1);
    if (w == 0)
        for (int c = lane; c < c0; c += 64) out[c] = P::finish(P::id(), divisor);

    WaveState st;
    st.ccur = c0;
    st.wbase = c0;
    fill_window(win, p, c0, ncol, lane);
    st.head_open = !PLANNED;   // (a planned chunk starts on a column start: its first column is its own)
    st.acc_in_lane0 = true;   // (both accumulators start as the identity in every lane)
    const int32_t p_c0 = __builtin_amdgcn_readfirstlane(win[0]);   // first element of column c0
    st.head_complete = p_c0 >= cs;
    st.has_next = c0 + 1 <= ncol;
    st.qnext = __builtin_amdgcn_readfirstlane(win[1]);
    RSP_STAMP(2);

    double acc0 = P::id(), acc1 = P::id();
    const int nbatches = (nrows + BATCH_ROWS - 1) / BATCH_ROWS;
    for (int b = 0; b < nbatches; ++b) {
#pragma unroll
        for (int g = 0; g < BATCH_ROWS / kGroupRows; ++g) {
            if (b == 0 && g == 1) RSP_STAMP(3);   // first group of 4 rows done
            if (b == 1 && g == 0) RSP_STAMP(4);   // first batch of 8 rows done
            const int row0 = b * BATCH_ROWS + g * kGroupRows;
            const int gs = cs + row0 * kRowElems;
            bool done = false;
            // the group's four rows after the per-element transform (free for plain sums)
            d2 t[kGroupRows];
            if (MASKED) {
                uint32_t m0[kGroupRows], m1[kGroupRows];
#pragma unroll
                for (int rr = 0; rr < kGroupRows; ++rr) {   // all eight probes in flight together
                    const int2 ij = iv[g * kGroupRows + rr];
                    if (LDSMAP) {   // (a row index outside the bitmap -- not a valid dgCMatrix -- reads word 0)
                        const uint32_t w0 = (uint32_t)ij.x >> 5, w1 = (uint32_t)ij.y >> 5;
                        m0[rr] = s_bitmap[w0 < (uint32_t)bitmap_words ? w0 : 0u];
                        m1[rr] = s_bitmap[w1 < (uint32_t)bitmap_words ? w1 : 0u];
                    } else {
                        m0[rr] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mr, (ij.x >> 5) * 4, 0, 0);
                        m1[rr] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mr, (ij.y >> 5) * 4, 0, 0);   // (sc0 / sc1 bits: same time; nt: 8.2 instead of 5.9 ms, profiles/r03_masked.md)
                    }
                }
#pragma unroll
                for (int rr = 0; rr < kGroupRows; ++rr) {
                    const int2 ij = iv[g * kGroupRows + rr];
                    const bool in0 = (m0[rr] >> (ij.x & 31)) & 1u, in1 = (m1[rr] >> (ij.y & 31)) & 1u;
                    const bool want = (OP == kOpMaskedIn);
                    t[rr].x = (in0 == want) ? v[g * kGroupRows + rr].x : 0.0;
                    t[rr].y = (in1 == want) ? v[g * kGroupRows + rr].y : 0.0;
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < kGroupRows; ++rr) {
                    t[rr].x = xf<OP>(v[g * kGroupRows + rr].x);
                    t[rr].y = xf<OP>(v[g * kGroupRows + rr].y);
                }
            }
            if (PLANNED && g == 0 && b == 0) {
                // what precedes the chunk's first column start belongs to the previous chunk (at most one group)
#pragma unroll
                for (int rr = 0; rr < kGroupRows; ++rr) {
                    const int e = gs + rr * kRowElems + 2 * lane;
                    if (e < xs0) t[rr].x = P::id();
                    if (e + 1 < xs0) t[rr].y = P::id();
                }
            }
            if (!P::kSum && ce - gs < kGroupElems) {   // (gs <= ce; written so that nothing overflows near 2^31)
                // max / min: the zero-filled lanes past the end of x (only the last, partial row
                // of the matrix has any) must not take part -- give them the identity
#pragma unroll
                for (int rr = 0; rr < kGroupRows; ++rr) {
                    const int e = gs + rr * kRowElems + 2 * lane;
                    if (e >= ce) t[rr].x = P::id();
                    if (e + 1 >= ce) t[rr].y = P::id();
                }
            }
            if (row0 < nrows) {
                // how many column ends fall inside this group of rows?
                // (only ends up to the chunk's own end count: later ones belong to other chunks)
                const uint32_t left = (uint32_t)(ce - gs);
                const uint32_t glim = left < (uint32_t)kGroupElems ? left : (uint32_t)kGroupElems;
                const uint32_t dq = (uint32_t)st.qnext - (uint32_t)gs;
                if (st.has_next && (dq - 1u) < glim) {
                    bool valid;
                    const int wq = load_next_ends(st, win, p, st.ccur + 1, ncol, lane, valid);
                    const uint32_t d = (uint32_t)wq - (uint32_t)gs;
                    const int n4 = __popcll(__ballot(valid && (d - 1u) < glim));
                    if (n4 >= kDenseMinEnds) {
                        // n4 saturates at 64 (one window read): count the rest of the group's ends
                        int nall = n4, last = n4;
                        for (int k = st.ccur + 1 + 64; last == 64 && nall < kDenseMaxEnds; k += 64) {
                            bool v2;
                            const int q2 = load_next_ends(st, win, p, k, ncol, lane, v2);
                            const uint32_t d2_ = (uint32_t)q2 - (uint32_t)gs;
                            last = __popcll(__ballot(v2 && (d2_ - 1u) < glim));
                            nall += last;
                        }
                        if (nall < kDenseMaxEnds)   // (beyond that: mostly empty columns, general path)
                            done = dense_group<P>(t, gs, glim, nall, lane, st, acc0, acc1, win, stage, p,
                                                      ncol, w, out, carry_head, divisor);
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < kGroupRows; ++rr) {
                const int r = g * kGroupRows + rr;
                const int row = row0 + rr;
                if (!done && row < nrows)
                    process_row<P>(t[rr].x, t[rr].y, cs + row * kRowElems, lane, st, acc0, acc1, win, hist, p,
                                       ncol, w, out, carry_head, divisor);
                v[r] = __builtin_bit_cast(
                    d2, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, (row + BATCH_ROWS) * 1024, AUX));
                if (MASKED)
                    iv[r] = __builtin_bit_cast(
                        int2, __builtin_amdgcn_raw_buffer_load_b64(ir, lane * 8, (row + BATCH_ROWS) * 512, AUX));
            }
        }
    }

    RSP_STAMP(5);
    if (PLANNED) {
        // The column still open at the grid end (if one reaches past it: ext > 0) is this chunk's: its running
        // result plus the tail's elements, one wave reduction.  Columns after it up to the next chunk's first
        // column are empty columns sitting exactly at that column's start: theirs is the identity.
        // (a chunk without a column start in its grid range, xs0 >= ce, owns nothing: the column crossing it is
        // finished by the chunk it starts in)
        if (ext > 0 && xs0 < ce) {
            double a = P::comb(acc0, acc1);
#pragma unroll
            for (int r = 0; r < kGroupRows; ++r) a = P::comb(a, P::comb(tail[r].x, tail[r].y));
            const double total = wave_allreduce<P>(a);
            if (lane == 0 && st.ccur < ncol) out[st.ccur] = P::finish(total, divisor);
            for (int c = st.ccur + 1 + lane; c < c_end; c += 64) out[c] = P::finish(P::id(), divisor);
        }
        return;
    }
    const double T = wave_allreduce<P>(P::comb(acc0, acc1));
    if (lane == 0) {
        if (st.head_open) {
            carry_head[w] = T;
            carry_tail[w] = P::id();
        } else {
            carry_tail[w] = T;
        }
        // everything the fix-up needs about this chunk in one 16-byte record: its first column, the
        // number of column ends inside, the chunk holding that column's first element and whether the
        // column starts exactly on that chunk's edge (then its first part is that chunk's head)
        // (0 <= ts <= w for a valid p[]; clamped so that an invalid one cannot send the fix-up outside
        // the carries)
        int32_t ts = cmap.chunk_of(p_c0 < 0 ? 0 : p_c0);
        ts = ts > w ? w : ts;
        carry_info[w] = make_int4(c0, st.ccur - c0, ts, (int64_t)p_c0 == cmap.start(ts) ? 1 : 0);
    }
    RSP_STAMP(6);
    }   // (w < nchunks)
    if (FOLD) {
        // Calls that are one round of waves (C2: 22 us in all): the fix-up as a second launch costs ~5 us, most of it the
        // gap between two dependent kernels.  Here the workgroup that finishes LAST runs it.  A device-scope fence per
        // wavefront (write back this XCD's L2) cost more than the launch it saves -- 111 us against 22.6, profiles/
        // r06_fold_fixup.md -- so the carries are made visible the cheap way: lane 0 stores its three records once more as
        // device-scope atomic stores (written through to memory, no cache maintenance), waits for them, the workgroup takes
        // a ticket, and the holder of the last ticket reads every record with device-scope loads.  The adds are those of
        // colsums_fixup_kernel in the same order (fixup_chunk): identical bits.  The ticket word belongs to the launching
        // stream (capi.hip) and is back at zero when the kernel ends.
        __shared__ int s_last;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (a record may have been stored by another lane of this wavefront)
        if (w < nchunks && lane == 0) {
            const double h = carry_head[w], t = carry_tail[w];
            const int4 inf = carry_info[w];
            __hip_atomic_store(&carry_head[w], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&carry_tail[w], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long* q = reinterpret_cast<unsigned long long*>(&carry_info[w]);
            __hip_atomic_store(&q[0], ((unsigned long long)(uint32_t)inf.y << 32) | (uint32_t)inf.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&q[1], ((unsigned long long)(uint32_t)inf.w << 32) | (uint32_t)inf.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the stores above have completed: s_waitcnt, no cache write-back)
        __syncthreads();
        if (threadIdx.x == 0)
            s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1 : 0;
        __syncthreads();
        if (s_last) {
            for (int base = 0; base < nchunks; base += WPG * 64)
                fixup_chunk<P, true>(base + (int)threadIdx.x, lane, ncol, nchunks, out, carry_head, carry_tail, carry_info, divisor);
            if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------
// fix-up: columns that cross chunk edges
// ---------------------------------------------------------------------------
// One thread per chunk w.  If a column that started in an earlier chunk ends in
// chunk w, its sum is   first + head[ts+1] + ... + head[w]   where ts is the chunk
// holding the column's first element and `first` is that chunk's tail (or its head
// when the column starts exactly on the chunk edge).  ts and the edge flag come from the
// main kernel's record, and the two partials of the usual case (a column spilling over
// ONE chunk edge) are loaded together with that record, so the launch is one memory
// round trip deep.  Short spans are added by the owning thread in ascending order; a span
// longer than 64 chunks (a giant column) is summed by the whole wavefront with a fixed
// lane-strided assignment and a fixed DPP tree, so one 1e9-long column costs
// microseconds instead of a serial walk.  Deterministic.
// COHERENT (the folded form: the records were written by other workgroups of the SAME launch): device-scope loads.
template <bool COHERENT>
__device__ __forceinline__ double carry_load(const double* q) {
    return COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
}
template <bool COHERENT>
__device__ __forceinline__ int4 carry_load(const int4* q) {
    if (!COHERENT) return *q;
    const unsigned long long* u = reinterpret_cast<const unsigned long long*>(q);
    const unsigned long long a = __hip_atomic_load(&u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(&u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_int4((int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)(uint32_t)b, (int)(uint32_t)(b >> 32));
}

template <class P, bool COHERENT>
__device__ __forceinline__ void fixup_chunk(int w, int lane, int32_t ncol, int32_t nchunks, double* __restrict__ out,
                                            const double* carry_head, const double* carry_tail, const int4* carry_info,
                                            double divisor) {
    // (called by whole wavefronts: the giant-column path below is wave-cooperative; lanes with w >= nchunks idle)
    bool need = false;
    int c = 0, ts = 0;
    double first = P::id(), head_w = P::id();
    if (w < nchunks) {
        const int4 inf = carry_load<COHERENT>(&carry_info[w]);                       // three independent loads
        head_w = carry_load<COHERENT>(&carry_head[w]);
        const double tail_prev = w > 0 ? carry_load<COHERENT>(&carry_tail[w - 1]) : P::id();
        c = inf.x;
        ts = inf.z;
        // a column ends in chunk w and it started in an earlier chunk
        need = inf.y != 0 && c < ncol && ts < w;
        if (need) first = inf.w ? carry_load<COHERENT>(&carry_head[ts]) : (ts == w - 1 ? tail_prev : carry_load<COHERENT>(&carry_tail[ts]));
    }
    const int span = w - ts;
    if (need && span <= 64) {
        double acc = first;
        for (int t = ts + 1; t < w; ++t) acc = P::comb(acc, carry_load<COHERENT>(&carry_head[t]));
        out[c] = P::finish(P::comb(acc, head_w), divisor);
    }
    // giant columns: whole wave per column, one after the other
    uint64_t m = __ballot(need && span > 64);
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int wl = __builtin_amdgcn_readlane(w, l);
        const int tl = __builtin_amdgcn_readlane(ts, l);
        double acc = P::id();
        // 8 independent loads in flight per lane; adds stay in ascending t order per lane
        for (int t0 = tl + 1 + lane; t0 <= wl; t0 += 64 * 8) {
            double h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 64 * u;
                h[u] = (t <= wl) ? carry_load<COHERENT>(&carry_head[t]) : P::id();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = P::comb(acc, h[u]);
        }
        const double total = P::comb(readlane_f64(first, l), wave_allreduce<P>(acc));
        if (lane == l) out[c] = P::finish(total, divisor);
    }
}

template <bool MEANS, int OP = kOpSum>
__global__ __launch_bounds__(256) void colsums_fixup_kernel(
    int32_t ncol, int32_t nchunks, double* __restrict__ out, const double* __restrict__ carry_head,
    const double* __restrict__ carry_tail, const int4* __restrict__ carry_info, double divisor,
    const int32_t* __restrict__ run_if = nullptr) {
    typedef Policy<MEANS, OP> P;
    if (run_if != nullptr && *run_if == 0) return;
    fixup_chunk<P, false>((int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(threadIdx.x & 63), ncol, nchunks, out, carry_head,
                   carry_tail, carry_info, divisor);
}

// n doubles from HBM into (page-locked, device-visible) host memory by a kernel: 16 bytes per lane, a wavefront writes 1 KiB of
// consecutive bytes per instruction, so the host link sees full-size writes.  The single-process multi-GPU call uses it for
// a shard's slice (multigpu.cpp RSP_GATHER_BLIT): behind the shard's kernels on the same stream it starts a few
// microseconds after them, where the runtime's copy command needs ~20 us before its first byte moves.
__global__ __launch_bounds__(256) void copy_f64_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
    const int64_t pairs = n >> 1;
    const d2* s2 = reinterpret_cast<const d2*>(src);
    d2* t2 = reinterpret_cast<d2*>(dst);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * blockDim.x) t2[k] = s2[k];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}
hipError_t launch_copy_f64(const double* d_src, double* dst, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if ((((uintptr_t)d_src | (uintptr_t)dst) & 15) != 0) {   // (a slice that starts on an odd column: the copy command does it)
        return hipMemcpyAsync(dst, d_src, (size_t)n * 8, hipMemcpyDeviceToHost, stream);
    }
    const int64_t want = (n / 2 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
    hipLaunchKernelGGL(copy_f64_kernel, dim3(blocks), dim3(256), 0, stream, d_src, dst, n);
    return hipGetLastError();
}

// ticket words of the folded fix-up (one per launching stream: capi.hip hands them out); zero between launches
constexpr int kFoldSlots = 64;
__device__ uint32_t g_fold_tickets[kFoldSlots];
hipError_t fold_ticket_address(int slot, uint32_t** out) {
    void* base = nullptr;
    const hipError_t e = hipGetSymbolAddress(&base, HIP_SYMBOL(g_fold_tickets));
    if (e != hipSuccess) return e;
    *out = (uint32_t*)base + (slot < 0 ? 0 : slot % kFoldSlots);
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// read-only ceiling (measurement helper behind rsp_debug_read_ceiling_device)
// ---------------------------------------------------------------------------
// What the memory system delivers to the main kernel's ACCESS SHAPE with all column work taken away: the same
// chunk grid (ChunkMap, one wavefront per chunk, 4 wavefronts per workgroup), the same buffer descriptor per
// chunk, the same 1 KiB `nt` loads in the same rolling register pipeline, two adds per lane and row -- no p[],
// no column ends, no result stores.  bench.py times it over the same x in the same run: the practical ceiling
// SURVEY.md 8(d) asks for next to the 8 TB/s spec peak.
template <int BATCH_ROWS>
__global__ __launch_bounds__(kWavesPerWG * 64) void read_ceiling_kernel(const double* __restrict__ x, int32_t nnz,
                                                                        ChunkMap cmap, int32_t nchunks,
                                                                        double* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * kWavesPerWG + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= nchunks) return;
    const int32_t cs = (int32_t)cmap.start(w);
    const int64_t ce64 = (int64_t)cs + cmap.elems(w);
    const int32_t ce = ce64 < (int64_t)nnz ? (int32_t)ce64 : nnz;
    const int32_t nrows = (int32_t)(((int64_t)ce - cs + 127) >> 7);
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + cs), 0, (int)((uint32_t)(ce - cs) * 8u), 0x00020000);
    const int voff = lane * 16;
    d2 v[BATCH_ROWS];
#pragma unroll
    for (int r = 0; r < BATCH_ROWS; ++r)
        v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, r * 1024, kLoadAux));
    double a0 = 0.0, a1 = 0.0;
    for (int b = 0; b < nrows; b += BATCH_ROWS) {
#pragma unroll
        for (int r = 0; r < BATCH_ROWS; ++r) {
            a0 += v[r].x;
            a1 += v[r].y;
            v[r] = __builtin_bit_cast(
                d2, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, (b + r + BATCH_ROWS) * 1024, kLoadAux));
        }
    }
    // keeps the loads alive; a lane whose sum is exactly this value (it will not be) stores it
    if (a0 + a1 == -0x1.23456789abcdep+1000) sink[0] = a0 + a1;
}

hipError_t launch_read_ceiling(const double* d_x, int32_t nnz, const LaunchPlan& plan, double* d_sink,
                               hipStream_t stream) {
    if (nnz <= 0) return hipSuccess;
    const ChunkMap cmap{plan.chunk_elems, plan.nbody, plan.tail_elems};
    const dim3 grid((plan.nchunks + kWavesPerWG - 1) / kWavesPerWG), block(kWavesPerWG * 64);
    if (plan.short_pipeline)
        hipLaunchKernelGGL((read_ceiling_kernel<4>), grid, block, 0, stream, d_x, nnz, cmap, plan.nchunks, d_sink);
    else
        hipLaunchKernelGGL((read_ceiling_kernel<kBatchRows>), grid, block, 0, stream, d_x, nnz, cmap, plan.nchunks,
                           d_sink);
    return hipGetLastError();
}

// nnz == 0: every column is empty (0 for sums, -inf / +inf for max / min)
__global__ void colsums_fill_kernel(double* __restrict__ out, int32_t ncol, double value) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncol) out[c] = value;
}

// nnz per column as doubles (the "count" reduction: no pass over x at all)
__global__ void colsums_count_kernel(const int32_t* __restrict__ p, double* __restrict__ out, int32_t ncol) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncol) out[c] = (double)(p[c + 1] - p[c]);
}

// ---------------------------------------------------------------------------
// synthetic values (bench / tests); mirrors oracle_gen_value bit for bit
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void gen_values_kernel(double* __restrict__ x, int64_t n, uint64_t seed,
                                  uint64_t first_idx, int kind) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const uint64_t h = mix64(seed * 0xD1342543DE82EF95ull + (first_idx + (uint64_t)k));
        double val;
        if (kind == 1) {
            val = (double)(h >> 11) * 0x1.0p-53;
        } else {
            const int s = (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) +
                          (int)((h >> 24) & 255);
            val = (double)(s - 510) / 100.0;
        }
        x[k] = val;
    }
}

// Row indices for synthetic matrices: the k entries of a column are drawn one per stratum,
// stratum r = [r * nrow / k, (r + 1) * nrow / k) in integer arithmetic, so rows are ascending
// and distinct (a valid dgCMatrix column) whenever k <= nrow.  One thread per column.
// Integer-only: oracle_gen_row_indices reproduces it bit for bit.
__global__ void gen_row_indices_kernel(int32_t* __restrict__ i, const int32_t* __restrict__ p,
                                       int32_t nrow, int32_t ncol, uint64_t seed) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int lo = p[c], k = p[c + 1] - p[c];
    for (int r = 0; r < k; ++r) {
        const uint64_t h = mix64(seed * 0xD1342543DE82EF95ull + 0x5bd1e995ull + (uint64_t)(lo + r));
        const uint64_t s0 = (uint64_t)r * (uint64_t)nrow / (uint64_t)k;
        const uint64_t s1 = (uint64_t)(r + 1) * (uint64_t)nrow / (uint64_t)k;
        const uint64_t width = s1 > s0 ? s1 - s0 : 1;
        uint64_t row = s0 + (((h >> 32) * width) >> 32);
        if (row >= (uint64_t)nrow) row = (uint64_t)nrow - 1;
        i[lo + r] = (int32_t)row;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_column_sums(const double* d_x, const int32_t* d_p, int32_t ncol, int32_t nnz,
                              double* d_out, const LaunchPlan& plan, void* d_workspace,
                              double divisor, bool means, hipStream_t stream, int op,
                              const int32_t* rows_i, const uint32_t* row_bitmap, int32_t bitmap_words,
                              const int2* plan_rec, const int32_t* run_if, uint32_t* fold_ticket) {
    if (ncol <= 0) return hipSuccess;
    if (op == kOpCount) {   // nnz per column: offsets only, x is not read
        hipLaunchKernelGGL(colsums_count_kernel, dim3((ncol + 255) / 256), dim3(256), 0, stream, d_p, d_out, ncol);
        return hipGetLastError();
    }
    if (nnz <= 0) {
        const double empty = op == kOpMax ? -__builtin_huge_val() : (op == kOpMin ? __builtin_huge_val() : 0.0);
        hipLaunchKernelGGL(colsums_fill_kernel, dim3((ncol + 255) / 256), dim3(256), 0, stream, d_out, ncol,
                           empty);
        return hipGetLastError();
    }
    char* ws = (char*)d_workspace;
    double* carry_head = (double*)ws;
    double* carry_tail = carry_head + plan.nchunks;
    int4* carry_info = (int4*)(carry_tail + plan.nchunks);   // (16-byte aligned: the workspace is, and 16 * nchunks bytes precede it)
    const ChunkMap cmap{plan.chunk_elems, plan.nbody, plan.tail_elems};
    const dim3 grid((plan.nchunks + kWavesPerWG - 1) / kWavesPerWG), block(kWavesPerWG * 64);
    // experiment ids >= 16: (id - 16) KiB of unused dynamic LDS per workgroup, which lowers the number of
    // resident waves per CU (occupancy sweeps, tools/taper_sweep.py)
    const unsigned extra_lds = plan.variant >= 16 ? (unsigned)(plan.variant - 16) * 1024u : 0u;
    if (plan_rec != nullptr && op == kOpSum && nnz > 0) {
        // inspector-executor form: ONE launch, no workspace (rsp_column_sums_planned_device)
#define RSP_LAUNCH_PLANNED(BR, MEANS_)                                                                          \
        hipLaunchKernelGGL((colsums_chunks_kernel<BR, MEANS_, kLoadAux, kWavesPerWG, kOpSum, false, true>), grid,   \
                           block, 0, stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out, (double*)nullptr,     \
                           (double*)nullptr, (int4*)nullptr, divisor, (const int32_t*)nullptr,                      \
                           (const uint32_t*)nullptr, 0, plan_rec)
        if (plan.short_pipeline) {
            if (means) RSP_LAUNCH_PLANNED(4, true); else RSP_LAUNCH_PLANNED(4, false);
        } else {
            if (means) RSP_LAUNCH_PLANNED(kBatchRows, true); else RSP_LAUNCH_PLANNED(kBatchRows, false);
        }
#undef RSP_LAUNCH_PLANNED
        return hipGetLastError();
    }
#define RSP_LAUNCH_K(KERNEL, BR, AUX_)                                                              \
    do {                                                                                           \
        if (means)                                                                                 \
            hipLaunchKernelGGL((KERNEL<BR, true, AUX_>), grid, block, extra_lds, stream, d_x, d_p, ncol,   \
                               nnz, cmap, plan.nchunks, d_out, carry_head, carry_tail, \
                               carry_info, divisor, rows_i, row_bitmap, bitmap_words);             \
        else                                                                                       \
            hipLaunchKernelGGL((KERNEL<BR, false, AUX_>), grid, block, extra_lds, stream, d_x, d_p, ncol,  \
                               nnz, cmap, plan.nchunks, d_out, carry_head, carry_tail, \
                               carry_info, divisor, rows_i, row_bitmap, bitmap_words);             \
    } while (0)
#define RSP_LAUNCH_W(WPG_)                                                                              \
    do {                                                                                               \
        const dim3 g2((plan.nchunks + (WPG_) - 1) / (WPG_)), b2((WPG_) * 64);                          \
        if (means)                                                                                     \
            hipLaunchKernelGGL((colsums_chunks_kernel<kBatchRows, true, kLoadAux, WPG_>), g2, b2, 0,   \
                               stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out,     \
                               carry_head, carry_tail, carry_info, divisor, rows_i, row_bitmap,        \
                               bitmap_words);                                                          \
        else                                                                                           \
            hipLaunchKernelGGL((colsums_chunks_kernel<kBatchRows, false, kLoadAux, WPG_>), g2, b2, 0,  \
                               stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out,     \
                               carry_head, carry_tail, carry_info, divisor, rows_i, row_bitmap,        \
                               bitmap_words);                                                          \
    } while (0)
#define RSP_LAUNCH_OP(OP_)                                                                             \
    hipLaunchKernelGGL((colsums_chunks_kernel<kBatchRows, false, kLoadAux, kWavesPerWG, OP_>), grid,   \
                       block, 0, stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out,   \
                       carry_head, carry_tail, carry_info, divisor, rows_i, row_bitmap, bitmap_words,  \
                       (const int2*)nullptr, run_if)
    // row-restricted sums with a bitmap of 16-128 KB (about 1e5-1e6 rows): bitmap in LDS, shared by as many
    // wavefronts as fit beside it (each brings 5.25 KB of its own): one workgroup of 16 / 8 / 6 / 4 per CU.
    // (With 4 the call is bound by its occupancy: 32 KB of loads in flight per CU, 4.7 TB/s at 1e6 rows.)
    const size_t bitmap_bytes = (size_t)bitmap_words * 4;
    constexpr size_t kLdsPerCu = 160 * 1024, kLdsPerWave = sizeof(double) * kStageSlots + sizeof(int32_t) * kPWin;
#define RSP_LAUNCH_LDSMAP(WPG_)                                                                                      \
    do {                                                                                                            \
        static DynamicLdsLimit lim_in_##WPG_, lim_out_##WPG_;                                                       \
        hipError_t ea = lim_in_##WPG_.ensure(                                                                       \
            (const void*)colsums_chunks_kernel<kBatchRows, false, kLoadAux, WPG_, kOpMaskedIn, true>,               \
            (int)(kLdsPerCu - (WPG_) * kLdsPerWave));                                                               \
        if (ea == hipSuccess)                                                                                       \
            ea = lim_out_##WPG_.ensure(                                                                             \
                (const void*)colsums_chunks_kernel<kBatchRows, false, kLoadAux, WPG_, kOpMaskedOut, true>,          \
                (int)(kLdsPerCu - (WPG_) * kLdsPerWave));                                                           \
        if (ea != hipSuccess) return ea;                                                                            \
        const dim3 g2((plan.nchunks + (WPG_) - 1) / (WPG_)), b2((WPG_) * 64);                                       \
        if (op == kOpMaskedIn)                                                                                      \
            hipLaunchKernelGGL((colsums_chunks_kernel<kBatchRows, false, kLoadAux, WPG_, kOpMaskedIn, true>), g2,   \
                               b2, bitmap_bytes, stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out,            \
                               carry_head, carry_tail, carry_info, divisor, rows_i, row_bitmap, bitmap_words);      \
        else                                                                                                        \
            hipLaunchKernelGGL((colsums_chunks_kernel<kBatchRows, false, kLoadAux, WPG_, kOpMaskedOut, true>), g2,  \
                               b2, bitmap_bytes, stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out,            \
                               carry_head, carry_tail, carry_info, divisor, rows_i, row_bitmap, bitmap_words);      \
    } while (0)
    if ((op == kOpMaskedIn || op == kOpMaskedOut) && bitmap_bytes > kLdsBitmapMinBytes &&
        bitmap_bytes <= kLdsBitmapMaxBytes) {
        if (bitmap_bytes + 16 * kLdsPerWave <= kLdsPerCu)
            RSP_LAUNCH_LDSMAP(16);
        else if (bitmap_bytes + 8 * kLdsPerWave <= kLdsPerCu)
            RSP_LAUNCH_LDSMAP(8);
        else if (bitmap_bytes + 6 * kLdsPerWave <= kLdsPerCu)
            RSP_LAUNCH_LDSMAP(6);
        else
            RSP_LAUNCH_LDSMAP(4);
    } else if (op == kOpSumSquares) {
        RSP_LAUNCH_OP(kOpSumSquares);
    } else if (op == kOpSumAbs) {
        RSP_LAUNCH_OP(kOpSumAbs);
    } else if (op == kOpMaskedIn) {
        RSP_LAUNCH_OP(kOpMaskedIn);
    } else if (op == kOpMaskedOut) {
        RSP_LAUNCH_OP(kOpMaskedOut);
    } else if (op == kOpMax) {
        RSP_LAUNCH_OP(kOpMax);
    } else if (op == kOpMin) {
        RSP_LAUNCH_OP(kOpMin);
    } else
    if (fold_ticket != nullptr && plan.short_pipeline && plan.variant == 0 && op == kOpSum) {
        // one launch: the last workgroup to finish runs the fix-up (FOLD)
        if (means)
            hipLaunchKernelGGL((colsums_chunks_kernel<4, true, kLoadAux, kWavesPerWG, kOpSum, false, false, true>), grid, block, 0,
                               stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out, carry_head, carry_tail, carry_info, divisor,
                               rows_i, row_bitmap, bitmap_words, (const int2*)nullptr, (const int32_t*)nullptr, fold_ticket);
        else
            hipLaunchKernelGGL((colsums_chunks_kernel<4, false, kLoadAux, kWavesPerWG, kOpSum, false, false, true>), grid, block, 0,
                               stream, d_x, d_p, ncol, nnz, cmap, plan.nchunks, d_out, carry_head, carry_tail, carry_info, divisor,
                               rows_i, row_bitmap, bitmap_words, (const int2*)nullptr, (const int32_t*)nullptr, fold_ticket);
        return hipGetLastError();
    } else
    switch (plan.variant) {   // 0 = production; the rest are A/B builds (rsp_set_experiment)
        case 1: RSP_LAUNCH_K(colsums_chunks_kernel, 16, kLoadAux); break;   // 16 rows in flight
        case 2: RSP_LAUNCH_W(1); break;                                     // 1 wavefront per workgroup
        case 3: RSP_LAUNCH_W(2); break;                                     // 2 wavefronts per workgroup
        case 4: RSP_LAUNCH_K(colsums_chunks_kernel, kBatchRows, 0); break;  // default cache policy
        case 5: RSP_LAUNCH_K(colsums_chunks_kernel, 4, kLoadAux); break;    // 4 rows in flight
        default:
            if (plan.short_pipeline)
                RSP_LAUNCH_K(colsums_chunks_kernel, 4, kLoadAux);
            else
                RSP_LAUNCH_K(colsums_chunks_kernel, kBatchRows, kLoadAux);
            break;
    }
#undef RSP_LAUNCH_K
#undef RSP_LAUNCH_OP
#undef RSP_LAUNCH_W
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 fgrid((plan.nchunks + 255) / 256), fblock(256);   // one thread per chunk
#define RSP_FIXUP(MEANS_, OP_)                                                                        \
    hipLaunchKernelGGL((colsums_fixup_kernel<MEANS_, OP_>), fgrid, fblock, 0, stream, ncol,            \
                       plan.nchunks, d_out, carry_head, carry_tail, carry_info, divisor, run_if)
    if (op == kOpMax)
        RSP_FIXUP(false, kOpMax);
    else if (op == kOpMin)
        RSP_FIXUP(false, kOpMin);
    else if (means)
        RSP_FIXUP(true, kOpSum);
    else
        RSP_FIXUP(false, kOpSum);
#undef RSP_FIXUP
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// lean planned kernel: short columns, everything a chunk needs requested at once
// ---------------------------------------------------------------------------
// Inspector-executor form for matrices whose columns are all short (<= kLeanMaxColumn entries; BASELINE
// config 2: ~10 per column).  The inspector (capi.hip, inspect_lean) does not only locate the chunks' first
// columns, it REWRITES the part of p[] a chunk needs in the form the executor wants: for chunk w (R rows
// of x on the 1 KiB grid) a header {c0, ncols} and the ncols + 1 column starts relative to the chunk's grid
// position as 16-bit numbers, at a fixed stride -- so the wavefront asks for its rows of x, its header and its
// offsets in the same instant (2 B per column instead of 4, no dependent round trip, p[] itself is never
// read), puts rows and offsets into LDS and hands out the columns to its lanes, 64 at a time.  A lane adds its
// column's entries from LDS in storage order from +0.0: EVERY column comes out bit-identical to the reference
// loop (src/example.cpp:28-30), also the one that reaches past the chunk's grid end (the chunk reads one row
// more: the inspector guarantees no column reaches further).  One launch, no carries, no workspace.
//
// VALIDATE (the plan-free entry's own plans, capi.hip auto_enqueue): the image was made from the p[] that stood at this
// address some calls ago, and nobody has promised that it still stands there.  Every lane therefore also loads ITS
// column's two offsets from the p[] of THIS call and compares them with the image's: equal -> the sum it made from LDS is
// the sum of exactly [p[c], p[c + 1]) and is stored; different -> the column is summed again straight from x[p[c] ..
// p[c + 1]) by the whole wavefront (any length, clamped to [0, nnz]: reads stay in bounds for any p[]) and *stale is
// set, which makes the host inspect again behind a later call.  Every column belongs to exactly one lane of one chunk
// (the image is a complete image of SOME offsets of the same ncol), so the result is right for ANY p[] -- the image only
// decides how fast.  Costs 4 B per column of extra reads, one round trip behind the header, off the sums' own path.
template <bool MEANS, int R, bool VALIDATE = false>   // R = rows of x per chunk
__global__ __launch_bounds__(kWavesPerWG * 64) void colsums_lean_kernel(
    const double* __restrict__ x, int32_t nnz, const int2* __restrict__ hdr, const uint32_t* __restrict__ offs,
    int32_t stride_dwords, int32_t nchunks, double* __restrict__ out, double divisor,
    const int32_t* __restrict__ p = nullptr, int32_t ncol = 0, int32_t* __restrict__ stale = nullptr) {
#pragma clang fp contract(off)
    typedef Policy<MEANS, kOpSum> P;
    constexpr int kElems = (R + 1) * kRowElems;   // the chunk's rows and the one after
    __shared__ __attribute__((aligned(16))) double s_stage[kWavesPerWG][kElems + 32];
    extern __shared__ uint32_t s_offs[];   // kWavesPerWG x stride_dwords (as many offsets as the fullest chunk has)
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // runs of kLeanXcdRun workgroups (64 chunks) per XCD: the result lines neighbouring chunks share meet in one L2
    // (1e9 entries in columns of ~10: 1.86 -> 1.51 ms; of ~30: 1.63 -> 1.45; C2 16.6 -> 16.2 us; profiles/r03_c2.md)
    const int w = xcd_runs<kLeanXcdRun>((int)blockIdx.x, (int)gridDim.x) * kWavesPerWG + wave_in_wg;
    if (w >= nchunks) return;
    const int32_t cs = w * (R * kRowElems);
    const int32_t left = nnz - cs;
    const int32_t avail = left < kElems ? left : kElems;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + cs), 0, avail * 8, 0x00020000);
    // (VALIDATE: the header goes first -- loads return in order, so it is there first -- and the offsets of this call's
    // p[] are requested as soon as it is, while the rows of x are still on their way: the check then waits for nothing
    // the sums do not wait for anyway)
    int2 h = make_int2(0, 0);
    if (VALIDATE) h = hdr[w];
    d2 v[R + 1];
#pragma unroll
    for (int r = 0; r <= R; ++r)
        v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16, r * 1024, kLoadAux));
    if (!VALIDATE) h = hdr[w];
    // the chunk's offsets: stride_dwords dwords (two 16-bit offsets each), the same count for every chunk
    const uint32_t* mine = offs + (size_t)w * (size_t)stride_dwords;
    uint32_t od[kLeanMaxOffsetDwords / 64];
#pragma unroll
    for (int t = 0; t < kLeanMaxOffsetDwords / 64; ++t)
        od[t] = (t * 64 + lane) < stride_dwords ? mine[t * 64 + lane] : 0u;
    int32_t plo0 = 0, phi0 = 0;   // VALIDATE: this call's offsets of the chunk's first 64 columns
    if (VALIDATE) {
        const int c0v = __builtin_amdgcn_readfirstlane(h.x), ncv = __builtin_amdgcn_readfirstlane(h.y);
        if (lane < ncv) {
            plo0 = p[c0v + lane];
            phi0 = p[c0v + lane + 1];
        }
    }
    double* stage = s_stage[wave_in_wg];
    uint32_t* so = s_offs + (size_t)wave_in_wg * stride_dwords;
#pragma unroll
    for (int t = 0; t < kLeanMaxOffsetDwords / 64; ++t)
        if (t * 64 + lane < stride_dwords) so[t * 64 + lane] = od[t];
#pragma unroll
    for (int r = 0; r <= R; ++r) *(d2*)&stage[r * kRowElems + 2 * lane] = v[r];
    __builtin_amdgcn_wave_barrier();
    const int c0 = __builtin_amdgcn_readfirstlane(h.x), ncols = __builtin_amdgcn_readfirstlane(h.y);
    const uint16_t* off16 = (const uint16_t*)so;
    for (int t0 = 0; t0 < ncols; t0 += 64) {
        const int col = t0 + lane;
        const bool active = col < ncols;
        int32_t plo = plo0, phi = phi0;
        if (VALIDATE && t0 > 0 && active) {   // (c0 + ncols <= ncol by construction of the image: p[c0 + col + 1] exists)
            plo = p[c0 + col];
            phi = p[c0 + col + 1];
        }
        const int lo = active ? (int)off16[col] : 0;
        const int hi = active ? (int)off16[col + 1] : 0;
        const int n = hi - lo;   // (0 <= lo <= hi <= kElems by construction of the plan)
        const double* sp = stage + lo;
        double s = 0.0;
        const int nquads = n >> 2;
        for (int q = 0; __ballot(q < nquads) != 0ull; ++q) {   // whole quads first, then the last 0-3 (as in dense_group)
            const double e0 = sp[4 * q], e1 = sp[4 * q + 1], e2 = sp[4 * q + 2], e3 = sp[4 * q + 3];
            const double t = (((s + e0) + e1) + e2) + e3;
            s = q < nquads ? t : s;
        }
        {
            const double* tp = sp + 4 * nquads;
            const int rem = n & 3;
            const double e0 = tp[0], e1 = tp[1], e2 = tp[2];
            const double t0_ = s + e0, t1 = t0_ + e1, t2 = t1 + e2;
            s = rem == 0 ? s : (rem == 1 ? t0_ : (rem == 2 ? t1 : t2));
        }
        if (!VALIDATE) {
            if (active) out[c0 + col] = P::finish(s, divisor);
        } else {
            // the image's [lo, hi) against this call's offsets (unsigned: cs + lo may pass 2^31 - 1 near the end of x)
            const bool same = (uint32_t)plo == (uint32_t)cs + (uint32_t)lo && (uint32_t)phi == (uint32_t)cs + (uint32_t)hi;
            if (active && same) out[c0 + col] = P::finish(s, divisor);
            uint64_t bad = __ballot(active && !same);
            if (bad != 0ull && lane == 0) *stale = 1;
            while (bad != 0ull) {   // rare: p[] changed under the plan -- these columns straight from x, whole wavefront each
                const int l = __builtin_ctzll(bad);
                bad &= bad - 1;
                int32_t a = __builtin_amdgcn_readlane(plo, l), b = __builtin_amdgcn_readlane(phi, l);
                a = a < 0 ? 0 : (a > nnz ? nnz : a);
                b = b < a ? a : (b > nnz ? nnz : b);
                // (compensated: a column of any length may stand here -- 1e9 entries over 64 lanes are 1.6e7 sequential adds
                // per lane, whose rounding alone would reach the 1e-12 bar; this path is rare, four flops per entry cost nothing)
                double part = 0.0, comp = 0.0;
                for (int64_t j = (int64_t)a + lane; j < (int64_t)b; j += 64) {
                    const double y = x[j] - comp, t = part + y;
                    comp = (t - part) - y;
                    if (!(__builtin_fabs(t) < __builtin_huge_val())) comp = 0.0;   // (an infinity or a NaN -- NA_real_ -- goes through as the plain += would carry it)
                    part = t;
                }
                const double total = wave_allreduce<P>(part);
                if (lane == l) out[c0 + col] = P::finish(total, divisor);
            }
        }
    }
}

hipError_t launch_column_sums_lean(const double* d_x, int32_t nnz, const int2* d_hdr, const uint32_t* d_offs,
                                   int32_t stride_dwords, int32_t nchunks, int32_t rows, double* d_out, double divisor,
                                   bool means, hipStream_t stream, const int32_t* d_p, int32_t ncol, int32_t* stale) {
    if (nchunks <= 0) return hipSuccess;
    const dim3 grid((nchunks + kWavesPerWG - 1) / kWavesPerWG), block(kWavesPerWG * 64);
    const size_t lds = (size_t)kWavesPerWG * (size_t)stride_dwords * 4;
    const bool validate = d_p != nullptr && stale != nullptr;
#define RSP_LEAN_V(R_, M_, V_)                                                                                       \
    hipLaunchKernelGGL((colsums_lean_kernel<M_, R_, V_>), grid, block, lds, stream, d_x, nnz, d_hdr, d_offs,          \
                       stride_dwords, nchunks, d_out, divisor, d_p, ncol, stale)
#define RSP_LEAN(R_)                                                                                               \
    do {                                                                                                            \
        if (means) {                                                                                                \
            if (validate) RSP_LEAN_V(R_, true, true);                                                               \
            else RSP_LEAN_V(R_, true, false);                                                                       \
        } else {                                                                                                    \
            if (validate) RSP_LEAN_V(R_, false, true);                                                              \
            else RSP_LEAN_V(R_, false, false);                                                                      \
        }                                                                                                           \
    } while (0)
    switch (rows) {
        case 2: RSP_LEAN(2); break;
        case 3: RSP_LEAN(3); break;
        case 4: RSP_LEAN(4); break;
        case 5: RSP_LEAN(5); break;
        case 6: RSP_LEAN(6); break;
        case 8: RSP_LEAN(8); break;
        case 12: RSP_LEAN(12); break;
        case 16: RSP_LEAN(16); break;
        default: return hipErrorInvalidValue;
    }
#undef RSP_LEAN
#undef RSP_LEAN_V
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// columns planned kernel: every column long -- one workgroup per column
// ---------------------------------------------------------------------------
// Inspector-executor form for matrices whose columns are ALL long and of similar length (>= kColumnsMinLen entries,
// none above four times the mean; the reference vignette's benchmark matrix: 1000 columns of ~1e4).  Such a matrix has
// no planned form on the chunk grid (every column crosses chunk edges by more than a group), and in the general form a
// call of its size pays a column search per chunk and a fix-up launch for ~1000 results.  Here the plan is only the
// inspector's KNOWLEDGE of the column lengths: workgroup c reads p[c], p[c + 1] (scalar) and its WPG wavefronts stream
// the column's rows of 128 entries round-robin, eight rows in flight each, through a buffer descriptor that ends
// where the column ends (what lies beyond reads as +0.0, so there is no tail code); lane sums -> fixed wave tree ->
// the wavefronts' sums added in wavefront order by one lane.  One launch, no workspace, no plan memory.
// Deterministic; like the general kernel on long columns, within tolerance of the reference's order, not its bits.
// GUARDED (the plan-free entry's own plans): the offsets are this call's, so the sums are right whatever stood at p[]
// when the plan was made; they are clamped to [0, nnz] (reads stay in bounds for any p[]) and a column outside
// [len_lo, len_hi] -- lengths the choice of this form did not rest on -- sets *stale, so that the host inspects again.
template <int WPG, bool MEANS, bool GUARDED = false>
__global__ __launch_bounds__(WPG * 64) void colsums_columns_kernel(const double* __restrict__ x,
                                                                   const int32_t* __restrict__ p, int32_t ncol,
                                                                   double* __restrict__ out, double divisor,
                                                                   int32_t nnz = 0, int32_t len_lo = 0, int32_t len_hi = 0,
                                                                   int32_t* __restrict__ stale = nullptr) {
    typedef Policy<MEANS, kOpSum> P;
    __shared__ double s_part[WPG];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = blockIdx.x;
    int32_t lo = p[c], hi = p[c + 1];
    if (GUARDED) {
        lo = lo < 0 ? 0 : (lo > nnz ? nnz : lo);
        hi = hi < lo ? lo : (hi > nnz ? nnz : hi);
        if (threadIdx.x == 0 && (hi - lo < len_lo || hi - lo > len_hi)) *stale = 1;
    }
    const int32_t n = hi - lo;
    const int32_t nrows = (int32_t)(((int64_t)n + kRowElems - 1) / kRowElems);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + lo), 0, n * 8, 0x00020000);
    double a0 = 0.0, a1 = 0.0;
    constexpr int kInFlight = 8;
    if (GUARDED && n > kColumnsMaxLen) {   // (a column no plan of this form was made for: its bytes may not fit a descriptor)
        double comp = 0.0;   // (compensated, as in the lean kernel's fall-back: any length may stand here)
        for (int64_t j = (int64_t)lo + threadIdx.x; j < (int64_t)hi; j += WPG * 64) {
            const double y = x[j] - comp, t = a0 + y;
            comp = (t - a0) - y;
            if (!(__builtin_fabs(t) < __builtin_huge_val())) comp = 0.0;
            a0 = t;
        }
    } else
    for (int r0 = wave; r0 < nrows; r0 += WPG * kInFlight) {
        d2 v[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u)   // (rows past the column's end lie outside the descriptor: they read as 0)
            v[u] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16, (r0 + u * WPG) * 1024, kLoadAux));
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            a0 += v[u].x;
            a1 += v[u].y;
        }
    }
    const double s = wave_allreduce<P>(a0 + a1);
    if (lane == 0) s_part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_part[0];
#pragma unroll
        for (int w = 1; w < WPG; ++w) t += s_part[w];
        out[c] = P::finish(t, divisor);
    }
}

hipError_t launch_column_sums_columns(const double* d_x, const int32_t* d_p, int32_t ncol, int32_t waves,
                                      double* d_out, double divisor, bool means, hipStream_t stream, int32_t nnz,
                                      int32_t len_lo, int32_t len_hi, int32_t* stale) {
    if (ncol <= 0) return hipSuccess;
    const bool guarded = stale != nullptr;
#define RSP_COLUMNS_G(W_, M_, G_)                                                                                 \
    hipLaunchKernelGGL((colsums_columns_kernel<W_, M_, G_>), dim3(ncol), dim3(W_ * 64), 0, stream, d_x, d_p, ncol,  \
                       d_out, divisor, nnz, len_lo, len_hi, stale)
#define RSP_COLUMNS(W_)                                                                                          \
    do {                                                                                                         \
        if (means) {                                                                                             \
            if (guarded) RSP_COLUMNS_G(W_, true, true);                                                          \
            else RSP_COLUMNS_G(W_, true, false);                                                                 \
        } else {                                                                                                 \
            if (guarded) RSP_COLUMNS_G(W_, false, true);                                                         \
            else RSP_COLUMNS_G(W_, false, false);                                                                \
        }                                                                                                        \
    } while (0)
    switch (waves) {
        case 2: RSP_COLUMNS(2); break;
        case 4: RSP_COLUMNS(4); break;
        case 8: RSP_COLUMNS(8); break;
        case 16: RSP_COLUMNS(16); break;
        default: return hipErrorInvalidValue;
    }
#undef RSP_COLUMNS
#undef RSP_COLUMNS_G
    return hipGetLastError();
}

hipError_t launch_gen_row_indices(int32_t* d_i, const int32_t* d_p, int32_t nrow, int32_t ncol,
                                  uint64_t seed, hipStream_t stream) {
    if (ncol <= 0) return hipSuccess;
    hipLaunchKernelGGL(gen_row_indices_kernel, dim3((ncol + 255) / 256), dim3(256), 0, stream, d_i, d_p, nrow,
                       ncol, seed);
    return hipGetLastError();
}

hipError_t launch_gen_values(double* d_x, int64_t n, uint64_t seed, uint64_t first_idx, int kind,
                             hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(gen_values_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_x, n, seed,
                       first_idx, kind);
    return hipGetLastError();
}

}  // namespace rsp

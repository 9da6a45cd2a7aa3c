// crossprod.hip -- Matrix::crossprod() on the device ("next" row f3 of SURVEY.md 8f).
//
// Reference inst/include/RcppSparse.h:159-194: dense ncol x ncol t(A) %*% A; every column
// pair is a sparse dot product by a sorted merge of the two row lists, res(c1,c2) +=
// x1 * x2 over the common rows in ascending row order (OpenMP over c1; the package's only
// parallel routine).  O(ncol^2) output: meant for matrices with few columns.
//
// Two kernels in the reference's accumulation order (bit-identical results), and two for the tall
// matrices the routine is meant for, where that order is a serial walk of every column (see "tall form"
// below: dense rank-k updates on the matrix cores, results within the floating-point tolerance --
// crossprod_tall_kernel up to 96 columns, crossprod_panels_kernel from 97 to 512: 8 / 12 / 16 column tiles with one
// workgroup per range of row panels, 24 / 32 tiles with three / four that share a range and one instantiation per real
// tile count):
//
//  * crossprod_rows_kernel (used when the caller provides a workspace).  The row-major form of
//    A is built first (integer row histogram, exclusive scan, cursor fill; the order of the
//    entries inside a row is irrelevant because a row contributes at most one product to any
//    output).  One workgroup of four wavefronts then owns a result column c1 -- or a slice of it, in
//    which case the row-major form is kept per slice: c1's entries are taken in ascending row order
//    and, for each (k, x1), x1 * x2 is added to acc[c2] for every stored (k, c2, x2) of row k, one
//    lane per entry of the row, the accumulators in LDS (three wavefronts stage the products, the
//    fourth adds them in order).  The work is exactly the sum over rows of nnz(row)^2 products;
//    nothing is spent on column pairs without common rows.
//
//  * crossprod_tiles_kernel (no workspace needed).  One workgroup owns a 64 x 64 tile of the
//    result (upper-triangular tile pairs only; the mirror image is written at the end).  It
//    walks the rows of A in blocks of 64: the 64 + 64 columns of the tile pair each keep a
//    cursor into their (ascending) row lists, the entries that fall into the current row block
//    are scattered into two dense LDS panels [row][column] with per-row presence masks, and
//    every thread accumulates its 4 x 4 sub-tile in registers over the rows that are present
//    on both sides.  Row blocks in which none of the 128 columns has an entry are skipped (the
//    next block starts at the smallest pending row).
//
// In these two, products are accumulated in ascending row order with a separate multiply and add (no
// FMA contraction), i.e. in the reference's order: results are bit-identical to the reference
// loop.  Only entries that are stored take part, so a non-finite value never meets a structural
// zero.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <stdint.h>
#include <type_traits>

#include "colsums_kernels.h"

namespace rsp {

constexpr int kXT = 64;   // tile edge (columns) and rows per block

__global__ __launch_bounds__(256) void crossprod_tiles_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, const int32_t* __restrict__ p,
    int32_t ncol, int32_t ntiles, double* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ double panel[2][kXT][kXT];                 // [side][row in block][column of tile]
    __shared__ unsigned long long present[2][kXT];        // [side][row]: bit c = column c stored
    __shared__ int s_min[4];

    // tile pair (I <= J) from the linear block index over the upper triangle
    int I = 0, rem = blockIdx.x;
    while (rem >= ntiles - I) {
        rem -= ntiles - I;
        ++I;
    }
    const int J = I + rem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = tid >> 4, tx = tid & 15;               // 4 x 4 sub-tile: rows 4*ty.., cols 4*tx..

    // threads 0..63 own the I-side columns, 64..127 the J-side columns
    const int side = (tid >> 6) & 1;
    const int slot = tid & 63;
    int cur = 0, end = 0;
    if (tid < 128) {
        const int c = (side == 0 ? I : J) * kXT + slot;
        if (c < ncol) {
            cur = p[c];
            end = p[c + 1];
        }
    }
    for (int k = tid; k < 2 * kXT * kXT; k += 256) (&panel[0][0][0])[k] = 0.0;
    if (tid < 2 * kXT) (&present[0][0])[tid] = 0ull;
    __syncthreads();

    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

    for (;;) {
        // smallest pending row over the 128 cursors -> start of the next row block
        int nxt = (tid < 128 && cur < end) ? ri[cur] : 0x7fffffff;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nxt = min(nxt, __shfl_xor(nxt, d, 64));
        if (lane == 0) s_min[wave] = nxt;
        __syncthreads();
        const int mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        if (mn == 0x7fffffff) break;                      // every column exhausted (uniform)
        const int r0 = mn & ~(kXT - 1);

        // scatter this block's entries into the panels
        const int first = cur;
        if (tid < 128) {
            while (cur < end) {
                const int k = ri[cur] - r0;
                if (k >= kXT) break;
                panel[side][k][slot] = x[cur];
                atomicOr(&present[side][k], 1ull << slot);
                ++cur;
            }
        }
        __syncthreads();

        // accumulate over the rows present on both sides, ascending
        for (int k = 0; k < kXT; ++k) {
            const unsigned long long mi = present[0][k], mj = present[1][k];
            if (mi == 0ull || mj == 0ull) continue;       // uniform
            const unsigned bi = (unsigned)(mi >> (4 * ty)) & 0xFu, bj = (unsigned)(mj >> (4 * tx)) & 0xFu;
            if (bi == 0u || bj == 0u) continue;
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = panel[0][k][4 * ty + q];
                b[q] = panel[1][k][4 * tx + q];
            }
#pragma unroll
            for (int qa = 0; qa < 4; ++qa)
#pragma unroll
                for (int qb = 0; qb < 4; ++qb)
                    if (((bi >> qa) & 1u) && ((bj >> qb) & 1u)) {
                        const double prod = a[qa] * b[qb];
                        acc[qa][qb] = acc[qa][qb] + prod;
                    }
        }
        __syncthreads();

        // clear what was written (cheaper than zeroing 64 KB per block)
        if (tid < 128)
            for (int q = first; q < cur; ++q) panel[side][ri[q] - r0][slot] = 0.0;
        if (tid < 2 * kXT) (&present[0][0])[tid] = 0ull;
        __syncthreads();
    }

    // write the tile and its mirror image (column-major ncol x ncol)
#pragma unroll
    for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int ci = I * kXT + 4 * ty + qa, cj = J * kXT + 4 * tx + qb;
            if (ci < ncol && cj < ncol) {
                out[(size_t)cj * ncol + ci] = acc[qa][qb];
                out[(size_t)ci * ncol + cj] = acc[qa][qb];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// row-major path
// ---------------------------------------------------------------------------------------------

constexpr int kXMaxWidth = 8192;  // accumulators (doubles) per wave in LDS

// tall form (few columns, long columns)
constexpr int kTallMaxCols = 512;        // 32 column tiles of 16 (above 16 tiles: the panel-table kernel only)
constexpr int kTallRows = 64;            // rows of A densified in LDS at a time (a "panel")
constexpr int kTallMaxGroups = 1280;     // workgroups = partial results to add up
constexpr int64_t kTallMinColumnLength = 4096;   // (average) below this the exact form's serial walk takes < 0.7 ms:
                                                  // bit-identical results are worth more there than the 1.5-2.5x the tall form gains

// The row-major form is kept per *slice* of result columns: "virtual row" k * nsplit + c / width
// holds the entries (c, x) of row k whose column lies in slice c / width, so that the wave that
// owns slice s of a result column reads exactly the entries it needs and nothing else.
// (Entries whose row index is outside [0, nrow) -- not a valid dgCMatrix -- are left out rather
// than allowed to address memory out of bounds.)
// In both passes a column is walked by gridDim.x wavefronts (64 entries each per step), the columns are spread
// over blockIdx.y and the 4 wavefronts of a block: long columns of a matrix with few of them still fill the chip.
__global__ __launch_bounds__(256) void xp_count_rows_kernel(const int32_t* __restrict__ ri,
                                                            const int32_t* __restrict__ p, int32_t nrow,
                                                            int32_t ncol, int32_t nsplit, int32_t width,
                                                            int32_t* __restrict__ cnt,
                                                            const int32_t* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;   // (tall form: the row-major form is only needed if x is not all finite)
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwaves = gridDim.y * 4;
    for (int c = wave; c < ncol; c += nwaves) {
        const int e1 = p[c + 1], slice = c / width;
        for (int64_t e = (int64_t)p[c] + (int64_t)blockIdx.x * 64 + lane; e < e1; e += (int64_t)gridDim.x * 64) {   // (64-bit: e1 may be 2^31 - 1)
            const int r = ri[e];
            if ((unsigned)r < (unsigned)nrow) atomicAdd(&cnt[(int64_t)r * nsplit + slice], 1);
        }
    }
}

// the counts start at zero -- as a kernel (not a memset) where the row-major form only stands by
__global__ __launch_bounds__(256) void xp_zero_if_kernel(int32_t* __restrict__ a, int64_t n,
                                                         const int32_t* __restrict__ run_if) {
    if (*run_if == 0) return;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) a[k] = 0;
}

// entry e of column c goes to the next free slot of its virtual row
__global__ __launch_bounds__(256) void xp_fill_rows_kernel(const double* __restrict__ x,
                                                           const int32_t* __restrict__ ri,
                                                           const int32_t* __restrict__ p, int32_t nrow,
                                                           int32_t ncol, int32_t nsplit, int32_t width,
                                                           int32_t* __restrict__ cursor, int32_t* __restrict__ rc,
                                                           double* __restrict__ rx,
                                                           const int32_t* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwaves = gridDim.y * 4;
    for (int c = wave; c < ncol; c += nwaves) {
        const int e1 = p[c + 1], slice = c / width;
        for (int64_t e = (int64_t)p[c] + (int64_t)blockIdx.x * 64 + lane; e < e1; e += (int64_t)gridDim.x * 64) {   // (64-bit: e1 may be 2^31 - 1)
            const int r = ri[e];
            if ((unsigned)r >= (unsigned)nrow) continue;
            const int pos = atomicAdd(&cursor[(int64_t)r * nsplit + slice], 1);
            rc[pos] = c;
            rx[pos] = x[e];
        }
    }
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// acc += v as one LDS instruction (ds_add_f64, an IEEE double add performed by the LDS unit).
// A wave's LDS instructions execute in issue order and the lanes of one instruction address
// distinct accumulators (a row stores a column once), so every accumulator still receives its
// products in ascending row order -- without the read / wait / add / write round trip.
__device__ __forceinline__ void lds_add_f64(double* a, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)a, v);
}

// workgroup barrier for LDS hand-offs only (no wait for outstanding global loads)
__device__ __forceinline__ void xr_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kXRound = 12;   // units a round of the row-major kernel hands from the stagers to the adder

// One workgroup of 4 wavefronts per (result column c1, slice).  A UNIT is (entry j of column c1, 64-wide
// segment g of that entry's row in this slice); the units in ascending (j, g) order are the reference's order
// of products for every output.  Round 2 split the work the way the rowSums accumulate pass does: wavefronts
// 1-3 STAGE units -- find them, load the row segment, multiply by x1 (a separate multiply, as in the
// reference) and put (byte offset of the accumulator, product) pairs into one of two LDS buffers, kXRound
// units a round -- and wavefront 0 ADDS the buffer staged in the previous round, unit by unit in order, with
// nothing to decide: ds_read_b32, ds_read_b64, ds_add_f64.  One wavefront doing both walked a column at
// ~270 ns per unit; the adder alone needs ~10 ns, so the column's chain is no longer what a call waits for.
// The headers of the next two groups of 64 entries (row, x1, extent of the row in the slice) travel while the
// current group is worked on.  LDS: span accumulators + 64 spare ones ("nothing to add" goes there) + 18 KB.
__global__ __launch_bounds__(256) void crossprod_rows_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, const int32_t* __restrict__ p,
    const int32_t* __restrict__ rp, const int32_t* __restrict__ rc, const double* __restrict__ rx,
    int32_t nrow, int32_t ncol, int32_t nsplit, int32_t width, double* __restrict__ out,
    const int32_t* __restrict__ run_if) {
#pragma clang fp contract(off)
    extern __shared__ double xr_sh[];
    double* acc = xr_sh;                                            // width accumulators + 64 spare slots
    double* bprod = acc + width + 64;                               // 2 x kXRound x 64 products
    int32_t* boff = (int32_t*)(bprod + 2 * kXRound * 64);           // 2 x kXRound x 64 byte offsets into acc
    if (run_if && *run_if == 0) return;                   // (stands by for the tall form, which then has run)
    const int c1 = blockIdx.x / nsplit, slice = blockIdx.x % nsplit, c_lo = slice * width;
    if (c1 >= ncol || c_lo >= ncol) return;               // (the launcher never creates such a slice)
    const unsigned span = (unsigned)(min(ncol - c_lo, width));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (unsigned c = tid; c < span + 64; c += 256) acc[c] = 0.0;
    __syncthreads();

    const int64_t e_beg = p[c1], e_end = p[c1 + 1];
    // header of a group of 64 entries of column c1, per lane: x1, first position and length of the entry's row
    // in this slice; stage A = (row index, x1), stage B = (position, length), each one group apart
    auto load_a = [&](int64_t e0, int& k, double& va) {
        const bool in = e0 + lane < e_end;
        k = in ? ri[e0 + lane] : -1;
        va = in ? x[e0 + lane] : 0.0;
    };
    auto load_b = [&](int k, int& rs, int& len) {
        rs = 0;
        len = 0;
        if ((unsigned)k < (unsigned)nrow) {
            const int64_t vr = (int64_t)k * nsplit + slice;   // this slice's part of row k
            rs = rp[vr];
            len = rp[vr + 1] - rs;
        }
    };
    int k1, k2, rs0, len0, rs1, len1;
    double va0, va1, va2;
    {
        int k0;
        load_a(e_beg, k0, va0);
        load_b(k0, rs0, len0);
        load_a(e_beg + 64, k1, va1);
    }
    for (int64_t e0 = e_beg; e0 < e_end; e0 += 64) {   // (64-bit: e_end may be 2^31 - 1)
        load_b(k1, rs1, len1);                 // group + 1: extents (its row indices arrived a group ago)
        load_a(e0 + 128, k2, va2);             // group + 2: row indices and x1
        // units of this group: lane j owns units [ustart, incl)
        const int nun = (len0 + 63) >> 6;
        int incl = nun;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const int ustart = incl - nun;
        const int U = __builtin_amdgcn_readlane(incl, 63);
        const int nrounds = (U + kXRound - 1) / kXRound;
        // Stagers fill round r while the adder drains round r - 1; the loads of round r + 1 are issued before
        // round r is written out, so one L2 round trip per GROUP is exposed, not one per round.  Two register
        // sets (even / odd rounds), hence the loop in steps of two.
        constexpr int Q = kXRound / 3;
        int cbA[Q], cbB[Q];
        double vbA[Q], vbB[Q], x1A[Q], x1B[Q];
        auto issue = [&](int r, int (&cb)[Q], double (&vb)[Q], double (&x1)[Q]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int u = r * kXRound + (wave - 1) + 3 * q;     // (uniform)
                cb[q] = c_lo + (int)span + lane;   // (a lane with nothing to add: a spare accumulator, product x1 * 0)
                vb[q] = 0.0;
                x1[q] = 0.0;
                if (u < U) {
                    const int j = __popcll(__ballot(incl <= u));    // the lane that owns unit u
                    const int g = u - __builtin_amdgcn_readlane(ustart, j);
                    const int rs_j = __builtin_amdgcn_readlane(rs0, j);
                    const int len_j = __builtin_amdgcn_readlane(len0, j);
                    x1[q] = readlane_f64(va0, j);
                    const int t = g * 64 + lane;
                    if (t < len_j) {
                        cb[q] = rc[rs_j + t];
                        vb[q] = rx[rs_j + t];
                    }
                }
            }
        };
        auto put = [&](int r, const int (&cb)[Q], const double (&vb)[Q], const double (&x1)[Q]) {
            int32_t* bo = boff + (r & 1) * kXRound * 64;
            double* bp = bprod + (r & 1) * kXRound * 64;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int s = (wave - 1) + 3 * q;
                // (every entry of the virtual row is in the slice, so c - c_lo < span; no decision left to make here)
                bo[s * 64 + lane] = (cb[q] - c_lo) * 8;
                bp[s * 64 + lane] = x1[q] * vb[q];
            }
        };
        auto drain = [&](int r) {
            const int32_t* bo = boff + (r & 1) * kXRound * 64;
            const double* bp = bprod + (r & 1) * kXRound * 64;
            int32_t off[kXRound];
            double pr[kXRound];
#pragma unroll
            for (int s = 0; s < kXRound; ++s) off[s] = bo[s * 64 + lane];
#pragma unroll
            for (int s = 0; s < kXRound; ++s) pr[s] = bp[s * 64 + lane];
#pragma unroll
            for (int s = 0; s < kXRound; ++s) lds_add_f64((double*)((char*)acc + off[s]), pr[s]);
        };
        if (wave != 0 && nrounds > 0) issue(0, cbA, vbA, x1A);
        for (int r = 0; r <= nrounds; r += 2) {   // iterations r and r + 1; every wavefront meets the same barriers
            if (wave != 0) {
                if (r + 1 < nrounds) issue(r + 1, cbB, vbB, x1B);
                if (r < nrounds) put(r, cbA, vbA, x1A);
            } else if (r >= 1) {
                drain(r - 1);
            }
            xr_lds_barrier();
            if (r + 1 <= nrounds) {
                if (wave != 0) {
                    if (r + 2 < nrounds) issue(r + 2, cbA, vbA, x1A);
                    if (r + 1 < nrounds) put(r + 1, cbB, vbB, x1B);
                } else {
                    drain(r);
                }
                xr_lds_barrier();
            }
        }
        va0 = va1; rs0 = rs1; len0 = len1;
        k1 = k2; va1 = va2;
    }
    __syncthreads();
    double* col = out + (size_t)c1 * ncol + c_lo;         // column c1 of the symmetric result
    for (unsigned c = tid; c < span; c += 256) col[c] = acc[c];
}

// ---------------------------------------------------------------------------------------------
// tall form: t(A) %*% A as a dense rank-k update on the matrix cores
// ---------------------------------------------------------------------------------------------
// The forms above give every output its products one after the other in ascending row order, which is
// what makes them bit-identical to the reference -- and what makes them slow on the shape crossprod is
// meant for: few columns, many rows.  48 columns of 4.5e7 rows are 48 serial walks of 4.5e7 steps:
// 24.8 s (round 2, 2^31 - 1 entries), the work itself being 1e11 multiply-adds.  For ncol <= 256 and
// columns of >= 4096 entries on average the library therefore sums in a different order: the rows of
// A are densified 64 at a time into an LDS panel P[64][ncol] (zero where nothing is stored) and
// C += t(P) P runs as v_mfma_f64_16x16x4_f64 over the 16 x 16 tile pairs I <= J, every workgroup over
// its own range of rows; the workgroups' results are added up in workgroup order.  Deterministic, within
// 1e-12 * sum |x1 x2| of the reference's order (tests/test_gpu_crossprod.py), not bit-identical.
// The same 48 x 4.5e7 matrix: 6.6 ms (the kernel reads the 26 GB of x and i once); 1e6 x 64 with 3.2e7 entries 0.26 ms against 89 ms; 64 columns of 4096 entries
// 0.26 ms against 0.68 ms (of 256 entries: 0.044 against 0.058 ms -- left to the bit-identical form).
// A product of a structural zero with a non-finite value would be NaN where the reference has nothing: the
// tall kernel looks at every value it loads and raises a flag if one is not finite; the combine kernel then
// leaves the output alone and the exact row-major path (whose kernels otherwise exit at once) does the work.
typedef double xp_v4f64 __attribute__((ext_vector_type(4)));
#ifndef RSP_TALL_SPLIT16
#define RSP_TALL_SPLIT16 2
#endif
#ifndef RSP_TALL_SPLIT12
#define RSP_TALL_SPLIT12 1
#endif
#ifndef RSP_TALL_XCD_PAIRS
#define RSP_TALL_XCD_PAIRS 1
#endif
constexpr int kTallSplit16 = RSP_TALL_SPLIT16, kTallSplit12 = RSP_TALL_SPLIT12;   // workgroups sharing a row range at 16 / 12 column tiles

// workgroup barrier for LDS hand-offs only (__syncthreads() would also wait for the loads just issued for the
// next panel: s_waitcnt vmcnt(0))
__device__ __forceinline__ void xp_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// tile pair number q of the upper triangle (I <= J) of an NT x NT grid of tiles, row by row
__device__ __forceinline__ void tall_pair(int q, int nt, int& I, int& J) {
    I = 0;
    while (q >= nt - I) {
        q -= nt - I;
        ++I;
    }
    J = I + q;
}

// One workgroup (NW = 4 wavefronts, 8 from 49 columns on, 16 from 129) per range of row panels, straight from the CSC
// arrays: a column's entries are in ascending row order, so the part of it that falls into the workgroup's
// rows is one contiguous piece (two binary searches per column at the start) and every panel takes the next
// few entries of each piece.
// NT = column tiles, CPW = 16 NT / NW.  Wavefront w loads columns w * CPW .. w * CPW + CPW - 1 (64 entries of
// each per panel, all loads issued before the first is used, the next panel's as soon as these sit in LDS)
// and owns the tile pairs q = w, w + NW, ... (at most MAXP), whose 16 x 16 accumulators stay in registers.
// Panels without entries are skipped: the next panel starts at the smallest row any column has pending.
// Measured on 48 columns x 4.5e7 rows (2^31 - 1 entries): 13.4 ms without the prefetch, 12.4 ms with it; panels of
// 128 rows (pieces of 512 B / 1 KB per column instead of 256 / 512 B, but two workgroups per CU instead of three)
// 14.1 ms; runs of 32 panels dealt round-robin, so that the resident workgroups read the same neighbourhood of
// every column, 15.5 ms.  What did help: a panel row stride of W + 1 doubles (a column's 64 rows otherwise sit in
// ONE LDS bank pair: 12.3 -> 10.7 ms) and telling the compiler to fit five workgroups per CU up to 48 columns (it
// used 130 registers where 96 do: 10.7 -> 7.1 ms = 3.7 TB/s).
// SPLIT (round 3; 2 from 193 columns on): the tile pairs of one range of rows are dealt to SPLIT workgroups,
// each of which densifies the panels for itself.  At 16 column tiles one workgroup of 16 wavefronts holds 9
// accumulator tiles per wavefront beside the 16 columns it loads -- more than the 128 registers a wavefront
// of a 1024-thread workgroup can have, and the spills made 1e6 x 256 cost 3.6 ms against 1.1 ms at 192 columns;
// with 5 tiles per wavefront nothing spills, at the price of reading x and i twice.
template <int NT, int NW, int SPLIT = 1>   // column tiles, wavefronts per workgroup, workgroups sharing a row range
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NT <= 3 || NW >= 8 ? 4 : 1, 8)))
void crossprod_tall_kernel(
    const double* __restrict__ x, const int32_t* __restrict__ ri, const int32_t* __restrict__ p, int32_t nrow,
    int32_t ncol, int64_t nnz, int32_t panels_per_group, int32_t* __restrict__ nonfinite,
    double* __restrict__ partial) {
    constexpr int W = NT * 16, NP = NT * (NT + 1) / 2, MAXP = (NP + NW * SPLIT - 1) / (NW * SPLIT), CPW = W / NW;
    __shared__ double panel[kTallRows][W + 1];   // (+1: the 64 rows of a column would otherwise sit in ONE LDS bank pair)
    __shared__ int32_t s_cur[W], s_end[W], s_next[NW];
    bool bad = false;   // a NaN or an infinity among the values this lane has loaded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ti[MAXP], tj[MAXP];
    xp_v4f64 acc[MAXP];
    // this wavefront's tile pairs: q = (s * NW + wave) * SPLIT + half
    // SPLIT workgroups share a range of rows and read the same pieces of x / i.  Workgroups are dealt to the 8 XCDs
    // round-robin, so blocks b and b + 8 meet in the same L2: in every run of 8 * SPLIT blocks, block 8 h + k is part h
    // of the run's k-th range -- the second reader of a piece finds it in its XCD's L2 instead of fetching it again.
    // (the last, incomplete run keeps neighbours together)
    int group = blockIdx.x / SPLIT, half = blockIdx.x - group * SPLIT;
    if (SPLIT > 1 && RSP_TALL_XCD_PAIRS) {
        const int run = blockIdx.x / (8 * SPLIT), in_run = blockIdx.x - run * (8 * SPLIT);
        if ((run + 1) * (8 * SPLIT) <= (int)gridDim.x) {
            group = run * 8 + (in_run & 7);
            half = in_run >> 3;
        }
    }
    auto pair_of = [&](int s) { return (s * NW + wave) * SPLIT + half; };
#pragma unroll
    for (int s = 0; s < MAXP; ++s) {
        tall_pair(pair_of(s) < NP ? pair_of(s) : 0, NT, ti[s], tj[s]);
        acc[s] = xp_v4f64{0.0, 0.0, 0.0, 0.0};
    }
    const int64_t R0 = (int64_t)group * panels_per_group * kTallRows;
    const int64_t R1 = R0 + (int64_t)panels_per_group * kTallRows < nrow ? R0 + (int64_t)panels_per_group * kTallRows : nrow;
    // this workgroup's piece of every column: [first entry with row >= R0, first entry with row >= R1)
    if (tid < W) {
        int32_t lo = 0, hi = 0;
        if (tid < ncol) {
            int64_t a = p[tid], b = p[tid + 1];
            a = a < 0 ? 0 : (a > nnz ? nnz : a);   // (an invalid p[] must not lead outside x / i)
            b = b < a ? a : (b > nnz ? nnz : b);
            int64_t l = a, h = b;
            while (l < h) {
                const int64_t m = (l + h) >> 1;
                if (ri[m] < R0) l = m + 1; else h = m;
            }
            lo = (int32_t)l;
            h = b;
            while (l < h) {
                const int64_t m = (l + h) >> 1;
                if (ri[m] < R1) l = m + 1; else h = m;
            }
            hi = (int32_t)l;
        }
        s_cur[tid] = lo;
        s_end[tid] = hi;
    }
    __syncthreads();
    // 64 entries of each of this wavefront's columns travel in registers; the next 64 are requested as soon as
    // these have gone into the panel (their count moves the cursor), so they arrive during the MFMA phase.
    // Cursors and ends are wave-uniform and live in scalar registers.
    // (16-wave workgroups, 12 or 16 columns per wavefront: the cursors stay in LDS -- only this wavefront touches
    // its columns' slots, and a wavefront's LDS operations execute in order -- because 32 of them in scalar
    // registers, plus the addresses made from them, spilled ~140 scalar registers into vector registers)
    constexpr bool kCursorsInLds = false;   // (tried for 16-wave workgroups: the loads then carry vector addresses, 1.9 instead of 1.1 ms at 192 columns)
    int32_t row[CPW], cur[kCursorsInLds ? 1 : CPW], end[kCursorsInLds ? 1 : CPW];
    double val[CPW];
    volatile int32_t* v_cur = s_cur + wave * CPW;
    volatile int32_t* v_end = s_end + wave * CPW;
    auto cursor = [&](int k) { return kCursorsInLds ? __builtin_amdgcn_readfirstlane(v_cur[k]) : cur[k]; };
    auto limit = [&](int k) { return kCursorsInLds ? __builtin_amdgcn_readfirstlane(v_end[k]) : end[k]; };
    auto fetch = [&]() {
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int32_t ck = cursor(k), ek = limit(k);
            const int32_t* rk = ri + ck;
            const double* xk = x + ck;
            const bool in = lane < ek - ck;
            row[k] = in ? rk[lane] : 0x7fffffff;
            val[k] = in ? xk[lane] : 0.0;
            // (This look at the value just requested makes the wavefront wait for it here.  Checking at the point of
            // use instead -- a true prefetch across the MFMA phase -- was measured in round 3 and LOSES: the 3 registers
            // per column then stay live through the MFMA phase and spill, 1e6 x 128 / 192 / 256: 0.73 -> 1.11,
            // 1.19 -> 1.67, 2.45 -> 4.14 ms.  Phase split at 256 columns, profiles/r03_crossprod.json: loads and cursor
            // bookkeeping alone 1.23 ms, the MFMA phase 1.2 ms on top, the LDS panel writes 0.1 ms.)
            bad |= ((uint32_t)__double2hiint(val[k]) & 0x7ff00000u) == 0x7ff00000u;
        }
    };
    if (!kCursorsInLds) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            cur[k] = __builtin_amdgcn_readfirstlane(s_cur[wave * CPW + k]);
            end[k] = __builtin_amdgcn_readfirstlane(s_end[wave * CPW + k]);
        }
    }
    // A batch of 64 entries per column travels in registers and is used up completely before the column is loaded
    // again (round 3: reloading every column from its cursor for every panel, as rounds 1-2 did, fetched every entry
    // about twice -- a 64-row panel takes ~32 of a column's 64 loaded entries at the densities this form is used at).
    // Consumed lanes hold row INT_MAX.  A column whose batch runs out inside a panel is refilled and scattered once more
    // in the same panel (rows of a column are distinct and ascending: a panel takes at most 64 of them, so one refill
    // per panel is enough).
    auto refill = [&](int k) {   // (nothing here looks at what it requests)
        const int32_t ck = cursor(k), ek = limit(k);
        const bool in = lane < ek - ck;
        row[k] = in ? (ri + ck)[lane] : 0x7fffffff;
        val[k] = in ? (x + ck)[lane] : 0.0;
    };
    auto take = [&](int k, int64_t r0_) {   // this panel's entries of column k out of its batch; true: batch used up, more to load
        const bool below = (int64_t)row[k] < r0_ + kTallRows;
        const uint32_t local = (uint32_t)((int64_t)row[k] - r0_);
        if (below) bad |= ((uint32_t)__double2hiint(val[k]) & 0x7ff00000u) == 0x7ff00000u;
        if (below && local < (uint32_t)kTallRows) panel[local][wave * CPW + k] = val[k];
        const int n = __popcll(__ballot(below));   // (also steps over rows below r0: an unsorted, invalid column)
        row[k] = below ? 0x7fffffff : row[k];
        cur[k] += n;
        return __ballot(row[k] != 0x7fffffff) == 0ull && cur[k] < end[k];
    };
    static_assert(!kCursorsInLds, "the batch form keeps its cursors in scalar registers");
    // Measured on one device, old against batch form, 1e6 rows: 16 / 48 / 64 / 96 columns 0.098 / 0.21 / 0.27 / 0.42 against
    // 0.11 / 0.22 / 0.30 / 0.46 ms (the refill inside the panel is an exposed wait that several resident workgroups used to
    // hide), 128 / 192 / 256 columns 0.71 / 1.15 / 2.42 against 0.69 / 1.13 / 2.30 ms: the batch form from 8 tiles on.
    constexpr bool kBatch = NT >= 8;
    // Round 4, 16 column tiles (one workgroup per CU: nobody else hides a wait): the batch is a RING.  Lane l of
    // column k always holds the entry e = l (mod 64) of the window [cur, cur + 64) of the column; the lanes a panel has
    // consumed at once request the entries 64 further on -- into their own registers, nothing new stays live -- and nobody
    // looks at them before the next panel's take, one MFMA phase later.  A window is full at the start of every take and a
    // 64-row panel takes at most 64 of a column's (distinct, ascending) rows, so there is never a second take inside a panel:
    // round 3's refill-and-take-again (an exposed memory round trip in about every other panel at 50 % density) is gone.
    // 1e6 x 256 / 200 / 192: profiles/r04_crossprod.json.
    constexpr bool kRing = NT >= 16;   // (12 tiles: 1.13 ms without the ring, 1.29 with it: its per-lane addresses cost more than its waits there)
    auto ring_load = [&](int k) {   // the first window of column k
        const uint32_t e = (uint32_t)cur[k] + (uint32_t)((lane - cur[k]) & 63);   // (unsigned: a cursor may stand just below 2^31)
        const bool in = e < (uint32_t)end[k];
        row[k] = in ? ri[e] : 0x7fffffff;
        val[k] = in ? x[e] : 0.0;
    };
    auto take_ring = [&](int k, int64_t r0_, int32_t& pending) {
        const bool below = (int64_t)row[k] < r0_ + kTallRows;
        const uint32_t local = (uint32_t)((int64_t)row[k] - r0_);
        if (below) bad |= ((uint32_t)__double2hiint(val[k]) & 0x7ff00000u) == 0x7ff00000u;
        if (below && local < (uint32_t)kTallRows) panel[local][wave * CPW + k] = val[k];
        const int n = __popcll(__ballot(below));   // (also steps over rows below r0: an unsorted, invalid column)
        // what this column still has to deliver: its oldest entry not taken; if the whole window went, the next panel may hold more
        const int32_t left = below ? 0x7fffffff : row[k];
        pending = left < pending ? left : pending;
        if (n == 64 && (int64_t)cur[k] + 64 < end[k]) {
            const int64_t np = r0_ + kTallRows;
            pending = np < (int64_t)pending ? (int32_t)np : pending;
        }
        // the consumed lanes' next entries (64 further on); past the column's end: INT_MAX, never below a panel's limit
        const uint32_t e = (uint32_t)cur[k] + (uint32_t)((lane - cur[k]) & 63) + 64u;   // (unsigned: no wrap below 2^32)
        const bool more = below && e < (uint32_t)end[k];
        if (below) row[k] = 0x7fffffff;
        if (more) {
            row[k] = ri[e];
            val[k] = x[e];
        }
        cur[k] += n;
    };
    if (kRing) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) ring_load(k);
    } else if (kBatch) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) refill(k);
    } else {
        fetch();
    }
    int64_t r0 = R0;
    while (r0 < R1) {
        int32_t pending = 0x7fffffff;   // smallest row this wavefront's columns still have to deliver
        if (kRing) {
            for (int k = tid; k < kTallRows * (W + 1); k += NW * 64) (&panel[0][0])[k] = 0.0;
            xp_lds_barrier();
#pragma unroll
            for (int k = 0; k < CPW; ++k) take_ring(k, r0, pending);
        } else if (kBatch) {
            for (int k = tid; k < kTallRows * (W + 1); k += NW * 64) (&panel[0][0])[k] = 0.0;
            xp_lds_barrier();
            uint32_t again = 0;   // columns refilled inside this panel (wave-uniform)
#pragma unroll
            for (int k = 0; k < CPW; ++k)
                if (take(k, r0)) again |= 1u << k;
#pragma unroll
            for (int k = 0; k < CPW; ++k)
                if (again & (1u << k)) refill(k);
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                if (again & (1u << k)) (void)take(k, r0);
                pending = row[k] < pending ? row[k] : pending;   // (consumed lanes hold INT_MAX)
            }
        } else {
            for (int k = tid; k < kTallRows * (W + 1); k += NW * 64) (&panel[0][0])[k] = 0.0;
            xp_lds_barrier();
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const bool below = (int64_t)row[k] < r0 + kTallRows;   // (lanes past the end hold INT_MAX)
                const uint32_t local = (uint32_t)((int64_t)row[k] - r0);
                if (below && local < (uint32_t)kTallRows) panel[local][wave * CPW + k] = val[k];
                const int n = __popcll(__ballot(below));   // (also steps over rows below r0: an unsorted, invalid column)
                // next row of this column: its first entry not taken, or unknown (then: the next panel) if all 64 were
                int32_t nx = below ? 0x7fffffff : row[k];
                const int32_t ck = cursor(k);
                if (n == 64 && (int64_t)ck + 64 < limit(k)) nx = (int32_t)(r0 + kTallRows < 0x7fffffff ? r0 + kTallRows : 0x7fffffff);
                pending = nx < pending ? nx : pending;
                if (kCursorsInLds) {
                    if (lane == 0) v_cur[k] = ck + n;
                } else {
                    cur[k] += n;
                }
            }
            fetch();
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int32_t o = __shfl_xor(pending, d, 64);
            pending = o < pending ? o : pending;
        }
        if (lane == 0) s_next[wave] = pending;
        xp_lds_barrier();
        // (operands of UNR k-steps are in registers at once: 2 x MAXP doubles each)
        constexpr int UNR = MAXP >= 5 ? 2 : 4;
#pragma unroll UNR
        for (int ks = 0; ks < kTallRows / 4; ++ks) {
            const double* prow = &panel[4 * ks + (lane >> 4)][lane & 15];
#pragma unroll
            for (int s = 0; s < MAXP; ++s)
                if (pair_of(s) < NP)   // (uniform per wavefront)
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(prow[16 * ti[s]], prow[16 * tj[s]], acc[s], 0, 0, 0);
        }
        int32_t nxt = s_next[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) nxt = s_next[w] < nxt ? s_next[w] : nxt;
        xp_lds_barrier();   // (panel and s_next are rewritten next)
        const int64_t step = r0 + kTallRows;
        const int64_t jump = R0 + (((int64_t)nxt - R0) / kTallRows) * kTallRows;   // the panel that holds row nxt
        r0 = nxt == 0x7fffffff ? R1 : (jump > step ? jump : step);
    }
    // (a structural zero has met a non-finite value in some product: the result below is not the reference's,
    // the combine kernel will leave it alone and the bit-identical kernels take over)
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(nonfinite, 1);
    // tile (I, J), element (row, col) = C(16 I + row, 16 J + col); lane: col = lane & 15, row = (lane >> 4) + 4 r
    double* mine = partial + (size_t)group * NP * 256;
#pragma unroll
    for (int s = 0; s < MAXP; ++s)
        if (pair_of(s) < NP) {
            double* t = mine + (size_t)pair_of(s) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[s][r];
        }
}

// ---------------------------------------------------------------------------------------------
// tall form, 16 column tiles (193-256 columns): panels found through a PANEL TABLE (round 4)
// ---------------------------------------------------------------------------------------------
// What bounds crossprod_tall_kernel at 16 tiles is not the matrix cores (busy 45 %) but the panel build: 256
// columns = 256 pieces per panel, each found by a dependent chain cursor -> address -> load -> ballot -> cursor
// (profiles/r04_crossprod.json: schedules of that chain were tried for two rounds).  This form takes the chain
// away instead.  A first pass over the row indices (4 bytes per entry, no atomics) writes a table
//     Ts[c][P], Te[c][P] = where column c's entries with a row in panel P (32 rows) begin and end
// -- at most 32 entries, rows ascend -- so that every address the matrix-core kernel will ever need is known
// before it starts.  That kernel: 8 wavefronts; a panel's cells of the tables travel two panels ahead, its entries
// (half a wavefront per column, 16 rounds) one panel ahead, in registers, while the current panel is multiplied;
// two panel buffers in LDS; no cursors, no ballots.  Its 136 tile pairs are dealt as whole tile ROWS (wavefront w:
// rows w and w + 8 of a circulant arrangement, 17 pairs), so the accumulators (136 registers) fit the 256 a
// wavefront of a 512-thread workgroup may have: nothing is densified twice (crossprod_tall_kernel's SPLIT is gone)
// and x / i are read once.  The workgroups' results are added up in a fixed order (crossprod_panels_combine_kernel),
// same tolerance as the tall form's, deterministic.
// 1e6 x 256, 50 % dense, one MI355X: table 0.17 ms + kernel 1.27 ms (MFMA pipes busy 70 %) + combine 0.02 ms: the
// call 1.53 ms against 2.10 ms (profiles/r04_crossprod.json; tools/compare_crossprod_panels.py).
// (Also built and measured: a panel-major COPY of the entries -- histogram, scan, fill with one atomic per run of
// neighbours -- read back as one contiguous piece per panel: the copy alone cost 1.15 ms at 1.28e8 entries.)
constexpr int kPanRows = 32;           // rows per panel of the table
// Row stride of a panel in LDS = W + 17 doubles (34 banks beyond a multiple of 64).  The MFMA operand reads take 16
// neighbouring doubles of 4 consecutive rows: with a stride of W + 1 (crossprod_tall_kernel's) the rows start 2 banks
// apart and two of them collide on 30 of their 32 banks (SQ_LDS_BANK_CONFLICT: 37 % of the LDS cycles); 34 banks apart
// they share 2.  The entries going in are the rows of ONE column: any odd stride spreads those over all the banks.
constexpr int kPanPad = 17;

// (a row outside the matrix -- negative: a large unsigned number -- lands in the last panel; the kernel drops the entry.
// Two instructions: the table pass is bound by its vector instructions, ~37 per entry at first, not by the 4 bytes it reads)
__device__ __forceinline__ int pan_of_row(int r, int shift, uint32_t last_panel) {
    const uint32_t q = (uint32_t)r >> shift;
    return (int)(q < last_panel ? q : last_panel);
}

// Ts[c][P] / Te[c][P] for every panel P and column c that meet; both are preset to 0 ("column c has nothing in
// panel P": an empty range).  An entry whose predecessor lies in another panel writes Ts, one whose
// successor does writes Te: at most two stores per entry, whatever the input; if the rows of a column do not
// ascend (not a valid dgCMatrix) several runs may claim a cell -- the cell then holds one of them, still indices of
// this column.
__global__ __launch_bounds__(256) void xp_panel_table_kernel(const int32_t* __restrict__ ri, const int32_t* __restrict__ p,
                                                             int32_t panel_shift, int32_t ncol, int64_t nnz, int64_t npanels,
                                                             int32_t* __restrict__ Ts, int32_t* __restrict__ Te) {   // (panels of 1 << panel_shift rows)
    // A lane takes four neighbouring entries (one aligned 16-byte load) and their two neighbours; two such steps are
    // in flight per wavefront.  (One entry per lane and step: 0.72 ms at 1.28e8 entries -- a wavefront waiting for
    // 256 bytes at a time; this form: profiles/r04_crossprod.json.)
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwaves = gridDim.y * 4;
    const uint32_t last_panel = (uint32_t)(npanels - 1);
    constexpr int UN = 2;   // (4: 0.20 instead of 0.17 ms)
    for (int c = wave; c < ncol; c += nwaves) {
        int64_t a = p[c], b = p[c + 1];
        a = a < 0 ? 0 : (a > nnz ? nnz : a);   // (an invalid p[] must not lead outside x / i)
        b = b < a ? a : (b > nnz ? nnz : b);
        int32_t* ts = Ts + (int64_t)c * npanels;
        int32_t* te = Te + (int64_t)c * npanels;
        const int64_t a4 = a & ~3ll;
        const int32_t a_rel = (int32_t)(a - a4), b_rel = (int32_t)(b - a4);
        for (int64_t g0 = a4 + (int64_t)blockIdx.x * (UN * 256); g0 < b; g0 += (int64_t)gridDim.x * (UN * 256)) {   // (wave-uniform)
            int4 v[UN];
            int32_t before[UN], after[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t e4 = g0 + u * 256 + lane * 4;
                v[u] = int4{0, 0, 0, 0};
                before[u] = after[u] = 0;
                if (e4 < b) {
                    if (e4 + 4 <= nnz) {
                        v[u] = *(const int4*)(ri + e4);
                    } else {   // (the last, partial quad of the array)
                        v[u].x = ri[e4];
                        if (e4 + 1 < nnz) v[u].y = ri[e4 + 1];
                        if (e4 + 2 < nnz) v[u].z = ri[e4 + 2];
                    }
                    if (e4 > 0) before[u] = ri[e4 - 1];
                    if (e4 + 4 < nnz) after[u] = ri[e4 + 4];   // (the neighbours' rows from the neighbouring lanes instead: 143 against 146 us, not kept)
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t e4 = g0 + u * 256 + lane * 4;
                const int32_t rr[6] = {before[u], v[u].x, v[u].y, v[u].z, v[u].w, after[u]};
                // (a store instruction costs the memory pipeline the same whether 4 or 64 of its lanes take part, and a piece
                // of a 50 % dense column is ~16 entries long: instead of one store per table and entry of the quad -- eight
                // instructions with a lane in sixteen active -- every lane queues its starts / ends and the wavefront stores
                // until no lane has one left: mostly once per table)
                int pan[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) pan[k] = pan_of_row(rr[k], panel_shift, last_panel);
                uint32_t starts = 0, ends = 0;   // bit k: entry e4 + k begins / ends its column's piece of a panel
                const int32_t rel = (int32_t)(e4 - a4);   // (32-bit from here: a column is shorter than 2^31)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t r = rel + k;
                    if (r < a_rel || r >= b_rel) continue;
                    if (r == a_rel || pan[k] != pan[k + 1]) starts |= 1u << k;
                    if (r == b_rel - 1 || pan[k + 2] != pan[k + 1]) ends |= 1u << k;
                }
                while (__ballot(starts != 0u) != 0ull) {   // (wave-uniform)
                    if (starts != 0u) {
                        const int k = __builtin_ctz(starts);
                        const int cur = k == 0 ? pan[1] : (k == 1 ? pan[2] : (k == 2 ? pan[3] : pan[4]));
                        ts[cur] = (int32_t)(e4 + k);
                        starts &= starts - 1;
                    }
                }
                while (__ballot(ends != 0u) != 0ull) {
                    if (ends != 0u) {
                        const int k = __builtin_ctz(ends);
                        const int cur = k == 0 ? pan[1] : (k == 1 ? pan[2] : (k == 2 ? pan[3] : pan[4]));
                        te[cur] = (int32_t)(e4 + k + 1);
                        ends &= ends - 1;
                    }
                }
            }
        }
    }
}

// has[P] = which workgroups of a panel range find entries in panel P, read off Te (an end of a non-empty piece is never
// 0): bit h = some column of the tiles that part h densifies holds entries there (one part = bit 0 below 24 tiles; at
// 24 / 32 tiles a workgroup densifies only the tiles its tile rows meet, panels_body).  A block looks at 256 panels x 32
// columns = two tiles: neighbouring threads, neighbouring panels.
// (Round 5, found by the soak: with ONE bit for all parts a panel could enter a workgroup's pipeline without an entry
// in any of ITS tiles, so nothing was noted in sSafe for it and the lanes without an entry went on reading entry 0 of
// the matrix -- whose row then lay in the panel being filled: x[0] in every column of that row, x[0]^2 added to column
// pairs that share no row.  Now a panel only enters where it holds entries of the workgroup's own tiles.)
__device__ __forceinline__ uint32_t xp_parts_of_tile(int tile, int rt, int split) {   // rt: the matrix's real column tiles (panels_body: RT)
    if (split <= 1) return 1u;
    const int nwr = (rt + split - 1) / split, need = nwr + rt / 2;
    uint32_t bits = 0;
    for (int h = 0; h < split; ++h) {
        int d = tile - h * nwr;
        if (d < 0) d += rt;
        if (d < need) bits |= 1u << h;
    }
    return bits;
}

__global__ __launch_bounds__(256) void xp_panel_has_kernel(const int32_t* __restrict__ Te, int32_t ncol, int64_t npanels,
                                                           int32_t /* padded tiles: not needed */, int32_t split,
                                                           uint32_t* __restrict__ has_words) {
    const int64_t P = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t bits = 0;
    if (P < npanels)
        for (int half = 0; half < 2; ++half) {
            const int c0 = blockIdx.y * 32 + 16 * half, c1 = c0 + 16 < ncol ? c0 + 16 : ncol;
            int any = 0;
            for (int c = c0; c < c1; ++c) any |= Te[(int64_t)c * npanels + P];
            if (any != 0) bits |= xp_parts_of_tile(c0 >> 4, (ncol + 15) >> 4, split);
        }
    // four panels' bytes share a word of has[]: the lanes of a quad put theirs together, one atomic per word
    uint32_t word = bits << (8 * (threadIdx.x & 3));
    word |= __shfl_xor(word, 1, 64);
    word |= __shfl_xor(word, 2, 64);
    if ((threadIdx.x & 3) == 0 && word != 0 && P < npanels) atomicOr(&has_words[P >> 2], word);
}

// One workgroup of NW wavefronts per range of panels.
// Tile pairs: tile row I meets the tiles (I + d) mod NT, d = 0 .. NT / 2 (rows below NT / 2) or d = 0 .. NT / 2 - 1 (the
// others) -- every unordered pair once.  16 tiles: 8 wavefronts, wavefront w has rows w and w + 8 (17 pairs, 136
// accumulator registers of the 256 it may have at two wavefronts per SIMD).  12 / 8 tiles: one row per wavefront (7 / 5
// pairs, the last one only for the rows below NT / 2), 12 wavefronts = three per SIMD / 8 wavefronts and two workgroups
// per CU.  The same code for every wavefront (only LDS offsets differ).  A tile that wraps (J < I) is the transpose of
// pair (J, I) and is stored so.
// Everything that is not an MFMA -- zeroing the other buffer, putting the next panel's entries into it, requesting
// the entries of the panel after that -- is dealt out over the groups of MFMAs of the current panel (panels_body).
typedef double xp_v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double xp_lds_f64;   // (LDS pointers as such: handed on as generic pointers they
typedef __attribute__((address_space(3))) int32_t xp_lds_i32;  // become FLAT accesses, which queue behind the global loads)
template <int NT, int NW, int PR, int SPLIT, bool WIDE, int MODE, int NAX>
__device__ __forceinline__ void panels_body(xp_lds_f64* __restrict__ panel, xp_lds_i32* __restrict__ sT,
                                            xp_lds_i32* __restrict__ sSafe, xp_lds_f64* __restrict__ sStray,
                                            const uint8_t* __restrict__ has,
                                            const double* __restrict__ x, const int32_t* __restrict__ ri,
                                            const int32_t* __restrict__ Ts, const int32_t* __restrict__ Te,
                                            int32_t ncol, int64_t npanels, int64_t P0, int64_t P1, int part,
                                            int32_t* __restrict__ nonfinite, double* __restrict__ mine) {
    // PR rows per panel (32; 16 at 32 tiles, where two buffers of 32 x 512 would not fit the LDS): a column's piece of a
    // panel is at most PR entries, PR lanes fetch it, CPI = 64 / PR columns go in one instruction.
    // SPLIT > 1 (24 / 32 tiles): the workgroups of a panel range each own NT / SPLIT tile rows, and tile row I only meets the
    // tiles I .. I + NT / 2 -- so a workgroup densifies only the WL = NT / SPLIT + NT / 2 tiles its rows meet (20 of 24, 24
    // of 32), kept in LDS in LOCAL order: local tile t = tile (t0 + t) mod NT with t0 its first tile row.  Round 5: the other
    // vector instructions beside the MFMAs were 3.4 per MFMA at these widths against 1.6 at 16 tiles, every workgroup
    // zeroing / requesting / scattering ALL columns for a third / a quarter of the MFMAs (profiles/r05_crossprod_wide.json).
    constexpr bool LOCAL = SPLIT > 1;
    constexpr int WL = LOCAL ? NT / SPLIT + NT / 2 : NT;   // tiles a workgroup densifies
    constexpr int W = WL * 16, W1 = W + kPanPad, NTH = NW * 64, CPI = 64 / PR, RND = W / (CPI * NW);
    constexpr bool TWO = NW * 2 == NT && SPLIT == 1;   // two tile rows per wavefront (16 tiles) or one
    static_assert(TWO || NW * SPLIT == NT, "a wavefront owns one or two whole tile rows; SPLIT workgroups share a range of panels");
    // NAX (24 / 32 tiles only, else NT / 2 + 1): pairs of a tile row when the matrix has FEWER column tiles than the kernel was
    // laid out for -- RT = ceil(ncol / 16) real tiles, NAX = RT / 2 + 1.  The tile rows then meet each other modulo RT, every
    // wavefront multiplies NAX pairs instead of NT / 2 + 1 (257 columns: 9 instead of 13), the workgroups of a panel range
    // share the RT real tile rows (NWR each; the wavefronts beyond multiply what they find and store nothing) and densify
    // NWR + RT / 2 tiles.  RT = NT: exactly the kernel of the full width.
    static_assert(NAX == NT / 2 + 1 || SPLIT > 1, "fewer pairs per tile row: the 24 / 32-tile forms only");
    constexpr int KS = PR / 4, NA = NAX;               // k-steps per panel; pairs of a tile row below RT / 2 (one more than of the others when RT is even)
    constexpr int NPW = TWO ? NT + 1 : NA;             // pairs per wavefront (one row: the last only if the row is below NT / 2)
    constexpr int kBufDoubles = PR * W1;
    constexpr int TC = (2 * W + NTH - 1) / NTH;        // cells of the tables per thread and panel (1; 2 at 32 tiles)
    static_assert(RND * CPI * NW == W && (PR == 32 || PR == 16), "every column in exactly one round");
    const int tid = threadIdx.x, lane = tid & 63, sub = lane / PR, l = lane % PR;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RT = LOCAL ? (ncol + 15) >> 4 : NT;      // real column tiles (LOCAL: the launcher picks NAX = RT / 2 + 1)
    const int NWR = LOCAL ? (RT + SPLIT - 1) / SPLIT : NW;   // tile rows per workgroup of a panel range
    const int trow = TWO ? wave : part * NWR + wave;   // this wavefront's (first) tile row
    const int t0 = LOCAL ? part * NWR : 0;             // (LOCAL: the workgroup's first tile row = local tile 0)
    const bool live = !LOCAL || (wave < NWR && trow < RT);   // (a wavefront without a tile row of its own stores nothing)
    const int need = NWR + RT / 2;                     // (LOCAL: local tiles 0 .. need - 1 are met by some tile row)
    static_assert(!LOCAL || !TWO, "local tile order is for one tile row per wavefront");
    xp_v4f64 acc[NPW];
#pragma unroll
    for (int s = 0; s < NPW; ++s) acc[s] = xp_v4f64{0.0, 0.0, 0.0, 0.0};
    // (wave-uniform) tile of pair s on the A side and on the B side; whether this wavefront has a pair s at all
    auto wrap = [&](int t) { return t >= RT ? t - RT : t; };
    auto tile_a = [&](int s) { return TWO && s >= NA ? trow + NW : trow; };
    auto tile_b = [&](int s) { return wrap(TWO && s >= NA ? trow + NW + (s - NA) : trow + s); };
    // where these tiles stand in the panel buffer (LOCAL: wave + s never reaches WL, nothing wraps)
    auto lds_tile_a = [&](int s) { return LOCAL ? wave : tile_a(s); };
    auto lds_tile_b = [&](int s) { return LOCAL ? wave + s : tile_b(s); };
    static_assert(!LOCAL || (NW - 1) + (NT / 2) < WL, "every pair's B tile lies inside the densified tiles");
    auto has_pair = [&](int s) { return TWO || s < NA - 1 || (RT & 1) != 0 || trow < RT / 2; };   // (RT odd: every row meets (RT + 1) / 2 tiles)

    auto zero_part = [&](int b, int m0, int m1) {   // 16-byte units tid + m * NTH, m0 <= m < m1
        auto* z = (__attribute__((address_space(3))) xp_v2f64*)(panel + b * kBufDoubles);
#pragma unroll
        for (int m = m0; m < m1; ++m) {
            const int k = tid + m * NTH;
            if (k < kBufDoubles / 2) z[k] = xp_v2f64{0.0, 0.0};
        }
    };
    constexpr int ZN = (kBufDoubles / 2 + NTH - 1) / NTH;   // (16 tiles: 9 -- eight full rounds and a tail)
    // In the loop no load is conditional and nothing is decided by what a load has just returned: a lane without an
    // entry of its own reads an entry of the PREVIOUS panel of the pipeline (one is noted whenever a panel's cells go
    // into sT), whose row the scatter then finds outside its panel like any stray row.  Only so does the compiler know
    // how many loads are outstanding when an entry is needed: with loads under the execution mask it branches around
    // them, no longer knows whether they were issued, and every round waits for the round before it -- s_waitcnt
    // vmcnt(1) where (30) would do, 1.66 ms at 1e6 x 256; with a select after the load ("row = lane has an entry ?
    // loaded : -1") the wavefront waits for what it has just requested (1.85 ms).
    int32_t treg[TC];
    auto global_col = [&](int col) {   // column of the matrix that stands at (local) column `col` of the panel buffer
        if (!LOCAL) return col;
        if ((col >> 4) >= need) return 0x7fffffff;     // (a tile no tile row of this workgroup meets: no column)
        const int g = col + 16 * t0;
        return g >= 16 * RT ? g - 16 * RT : g;
    };
    auto cell_of = [&](int k, bool& have, int& col, const int32_t*& tab) {   // cell tid + k NTH of {Ts[0..W), Te[0..W)}
        const int idx = tid + k * NTH;
        have = idx < 2 * W;
        col = idx < W ? idx : idx - W;
        tab = idx >= W ? Te : Ts;
    };
    auto load_T = [&](int64_t P) {   // row P of the two tables ([column][panel]), TC cells per thread (cell 0 when there is none)
#pragma unroll
        for (int k = 0; k < TC; ++k) {
            bool have; int col; const int32_t* tab;
            cell_of(k, have, col, tab);
            const int gcol = global_col(col);
            treg[k] = tab[have && gcol < ncol && P < P1 ? (int64_t)gcol * npanels + P : 0];
        }
    };
    auto put_T = [&](int q, int64_t P) {   // ... of panel P: zeros past the last panel and the last column; notes an entry of P in sSafe[q]
#pragma unroll
        for (int k = 0; k < TC; ++k) {
            bool have; int col; const int32_t* tab;
            cell_of(k, have, col, tab);
            const int32_t t = have && global_col(col) < ncol && P < P1 ? treg[k] : 0;
            if (have) sT[tid + k * NTH] = t;
            const unsigned long long ends = __ballot(tid + k * NTH >= W && t > 0);   // (cells from W on hold Te: the end of a piece that is not empty)
            if (ends != 0ull) {
                const int32_t last = __builtin_amdgcn_readlane(t, __builtin_ctzll(ends)) - 1;
                if (lane == 0) sSafe[q] = last;
            }
        }
    };
    int32_t r_[RND];
    double v_[RND];
    const char* ri_b = (const char*)ri;
    const char* x_b = (const char*)x;
    auto fetch = [&](int j, uint32_t at) {
        if (WIDE) {   // (byte offsets beyond 32 bits: 2^29 entries or more)
            r_[j] = ri[at];
            v_[j] = x[at];
        } else {      // (a scalar base and a 32-bit offset per lane: one instruction per address)
            r_[j] = *(const int32_t*)(ri_b + (uint64_t)(at << 2));
            v_[j] = *(const double*)(x_b + (uint64_t)(at << 3));
        }
    };
    // round j of the panel whose cells stand in sT: PR lanes per column (0, 0: nothing of this column; at most PR rows
    // of a valid column fall into a panel)
    const xp_lds_i32* const sTw = sT + wave * CPI + sub;
    auto request = [&](int j, int32_t safe) {
        const int32_t s = sTw[j * NW * CPI], n = sTw[W + j * NW * CPI] - s;
        fetch(j, l < n ? (uint32_t)(s + l) : (uint32_t)safe);
    };
    auto request_first = [&](int j) {   // (the first panel of a workgroup has no predecessor)
        const int32_t s = sTw[j * NW * CPI], n = sTw[W + j * NW * CPI] - s;
        r_[j] = -1;
        if (l < n) fetch(j, (uint32_t)(s + l));
    };
    // (LDS addresses as 32-bit numbers, the round's share of them a constant of the instruction: written as an index
    // into panel[] the compiler kept sixteen 64-bit constants, one per round, in 32 registers)
    const uint32_t cell0 = (uint32_t)(uintptr_t)panel + (uint32_t)(wave * CPI + sub) * 8u;
    const uint32_t stray0 = (uint32_t)(uintptr_t)sStray + (uint32_t)lane * 8u;   // (a cell per lane rather than one for all; the LDS bank conflicts the counters show, 36 % of its cycles, are the same either way)
    auto scatter = [&](int j, int b, int32_t r0) {
        const uint32_t local = (uint32_t)(r_[j] - r0);   // (-1, or a row of another panel: not below PR)
        const uint32_t at = local < (uint32_t)PR ? cell0 + (uint32_t)(b * kBufDoubles * 8) + __umul24(local, (uint32_t)(W1 * 8))
                                                        : stray0;
        ((xp_lds_f64*)(uintptr_t)at)[j * NW * CPI] = v_[j];
    };
    // The MFMAs of a panel (136 at 16 tiles) in groups of four (two); a group's B operands (and the two A operands of a k-step that begins
    // in it) are read from LDS while the group before it is multiplied.
    constexpr int NM = KS * NPW, G = NT >= 16 ? 4 : 2, NG = (NM + G - 1) / G;   // (12 / 8 tiles: 0.78 / 0.39 ms with groups of two, 0.81 / 0.40 with four)
    static_assert(2 * G <= NPW + 1, "a group and the one read ahead of it touch at most two k-steps: two sets of A operands");   // (16 tiles, groups of six: 12 more registers live, 54 spilled)
    double opb[2][G], opa[2][2];
    // (a pair's LDS address without the k-step's share, which is a constant of the instruction: made once per panel --
    // made per operand they were 150 vector instructions per panel, and vector instructions do not overlap f64 MFMAs)
    uint32_t opat[NPW], opat_a[2];
    const uint32_t lane_at = (uint32_t)(uintptr_t)panel + (uint32_t)((lane >> 4) * W1 + (lane & 15)) * 8u;
    auto operand_addresses = [&](int b) {
        const uint32_t base = lane_at + (uint32_t)(b * kBufDoubles * 8);
#pragma unroll
        for (int s = 0; s < NPW; ++s) opat[s] = base + (uint32_t)(128 * lds_tile_b(s));
        opat_a[0] = base + (uint32_t)(128 * lds_tile_a(0));
        opat_a[1] = base + (uint32_t)(128 * (TWO ? trow + NW : lds_tile_a(0)));
    };
    auto load_group = [&](int g) {
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int idx = g * G + t, ks = idx / NPW, s = idx % NPW;
            if (idx < NM) {
                opb[g & 1][t] = ((const xp_lds_f64*)(uintptr_t)opat[s])[ks * 4 * W1];
                if (s == 0) {
                    opa[ks & 1][0] = ((const xp_lds_f64*)(uintptr_t)opat_a[0])[ks * 4 * W1];
                    if (TWO) opa[ks & 1][1] = ((const xp_lds_f64*)(uintptr_t)opat_a[1])[ks * 4 * W1];
                }
            }
        }
    };
    auto mfma_group = [&](int g) {
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int idx = g * G + t, ks = idx / NPW, s = idx % NPW;
            if (idx < NM && has_pair(s))
                acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[ks & 1][TWO && s >= NA ? 1 : 0], opb[g & 1][t], acc[s], 0, 0, 0);
        }
    };

    // Only panels that hold entries enter the pipeline (has[], written with the tables).  "The next one" is asked for a
    // phase ahead with a vector load (hq), so that the common case -- the next panel holds entries too -- never waits;
    // a gap is walked with one exposed load per empty panel, a tenth of what multiplying it would cost.
    // (has[P]: bit h = panel P holds entries of the tiles part h densifies; one part: bit 0)
    auto holds = [&](int64_t P) { return ((int32_t)has[P] >> (LOCAL ? part : 0)) & 1; };
    auto next_panel = [&](int64_t P) {   // the first panel at or after P that holds entries (wave-uniform; P1 if none)
        while (P < P1 && __builtin_amdgcn_readfirstlane(holds(P)) == 0) ++P;
        return P;
    };
    // ---- the first panel into buffer 0, the second one requested, the third one's cells of the tables requested
    // (sSafe: entry 0 until a panel has been seen.  A range with ONE panel that holds entries never notes a second one,
    // and the dummy loads of its last phase would go wherever the LDS's previous owner left them pointing: a memory
    // fault on some runs of the edge sweep, 250 000 rows = 7813 panels = 252 ranges of 31 and one of 1.)
    if (tid < 2) sSafe[tid] = 0;
    __syncthreads();          // (before put_T notes a real one)
    int64_t Pc = next_panel(P0);
    zero_part(0, 0, ZN);
    load_T(Pc);
    put_T(0, Pc);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RND; ++j) request_first(j);
    int64_t Pn = Pc < P1 ? next_panel(Pc + 1) : P1;
    load_T(Pn);
#pragma unroll
    for (int j = 0; j < RND; ++j) scatter(j, 0, (int32_t)((Pc < P1 ? Pc : 0) * PR));
    __syncthreads();          // (everybody has read sT)
    put_T(1, Pn);
    __syncthreads();
    {
        const int32_t safe = __builtin_amdgcn_readfirstlane(sSafe[0]);   // an entry of panel Pc
#pragma unroll
        for (int j = 0; j < RND; ++j) request(j, Pc < P1 ? safe : 0);
    }
    int64_t Pnn = Pn < P1 ? next_panel(Pn + 1) : P1;
    load_T(Pnn);
    int32_t hq = Pnn + 1 < P1 ? (int32_t)has[Pnn + 1] : 0xff;   // (the byte as loaded: looked at a phase later)
    xp_lds_barrier();         // (everybody has read sT once more)
    int b = 0, q = 0;         // q: where this phase notes an entry of the panel whose cells it puts into sT (the other one: of the panel before)
    while (Pc < P1) {
        // buffer b holds panel Pc; the registers hold the entries of panel Pn and the cells of panel Pnn (requested a
        // phase ago); the other buffer still holds the panel before Pc
        const int o = b ^ 1;
        put_T(q, Pnn);        // (sT and sSafe[q] were last read before the barrier that ended the previous phase)
        int64_t Pnnn = Pnn + 1 < P1 ? Pnn + 1 : P1;
        if (Pnnn < P1 && ((__builtin_amdgcn_readfirstlane(hq) >> (LOCAL ? part : 0)) & 1) == 0) Pnnn = next_panel(Pnnn + 1);
        load_T(Pnnn);
        hq = Pnnn + 1 < P1 ? (int32_t)has[Pnnn + 1] : 0xff;
        const int32_t r0n = Pn < P1 ? (int32_t)(Pn * PR) : -2 * PR;   // (no next panel: no row is within PR of that)
        // Group g: its MFMAs, and a share of everything else -- the groups of the first two k-steps zero the other
        // buffer, 16 later ones move one round of entries each.  (sched_barrier: the compiler keeps this order.  All of
        // it in two blocks, before the first and after the last MFMA: 1.87 instead of 1.62 ms at 1e6 x 256; f64 MFMAs and
        // other vector instructions of a SIMD do not overlap -- tools/microbench/mfma_shadow.hip: 3-4 cycles per 32-bit
        // instruction on top of the MFMA's 64 -- but LDS writes, branches and waiting for LDS do.)
        constexpr bool mul = !(MODE & 1), side = !(MODE & 2);
        constexpr int ZG = (2 * NPW + G - 1) / G;   // the groups of the first two k-steps share the zeroing
        constexpr int RG = (RND + (NG - ZG - 1) - 1) / (NG - ZG - 1);   // rounds of entries per group after the barrier
        static_assert(NG > ZG + 1 && RG >= 1, "shares of the groups");
        int32_t safe = 0;
        if (mul) {
            operand_addresses(b);
            load_group(0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (mul && g + 1 < NG) load_group(g + 1);
            if (mul) mfma_group(g);
            if (g < ZG) zero_part(o, g * ZN / ZG, (g + 1) * ZN / ZG);
            if (g == ZG) {
                xp_lds_barrier();   // the other buffer is all zero (and sT is panel Pnn's) before anybody scatters (requests)
                safe = __builtin_amdgcn_readfirstlane(sSafe[q ^ 1]);   // an entry of panel Pn, for the lanes that have none of Pnn
            }
            if (g > ZG && side) {
#pragma unroll
                for (int k = 0; k < RG; ++k) {
                    const int j = (g - (ZG + 1)) * RG + k;
                    if (j < RND) {
                        scatter(j, o, r0n);
                        request(j, safe);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        xp_lds_barrier();     // everybody has multiplied panel Pc; the next one stands in the other buffer; sT has been read
        b = o;
        q ^= 1;
        Pc = Pn;
        Pn = Pnn;
        Pnn = Pnnn;
    }
    // A structural zero that meets a non-finite value makes a NaN here where the reference has nothing, and a NaN
    // stays: any sum that is not finite sends the call to the bit-identical kernels (the combine kernel then leaves
    // the result alone), as crossprod_tall_kernel does by looking at every value it loads -- here it costs 68
    // looks per lane and call instead of three instructions per entry.
    bool bad = false;
#pragma unroll
    for (int s = 0; s < NPW; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= ((uint32_t)__double2hiint(acc[s][r]) & 0x7ff00000u) == 0x7ff00000u;
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(nonfinite, 1);
    // pair (I <= J) number q; lane: element (row, col) of the tile A_side x B_side: col = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
    for (int s = 0; s < NPW; ++s) {
        if (!has_pair(s) || !live) continue;
        const int ta = tile_a(s), tb = tile_b(s);
        const bool wraps = tb < ta;              // computed C(ta, tb) = transpose of pair (tb, ta)
        const int I = wraps ? tb : ta, J = wraps ? ta : tb;
        const int q = I * NT - I * (I - 1) / 2 + (J - I);
        double* t = mine + (size_t)q * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            t[wraps ? col * 16 + row : row * 16 + col] = acc[s][r];
        }
    }
}

template <int NT, int NW, int PR, int SPLIT, bool WIDE, int MODE = 0, int NAX = NT / 2 + 1>   // (MODE, measurements only: 1 = no MFMAs, 2 = no entries moved; NAX: panels_body)
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NT >= 16 ? 2 : (NT == 12 ? 3 : 4), NT >= 16 ? 2 : (NT == 12 ? 3 : 4))))
void crossprod_panels_kernel(const double* __restrict__ x, const int32_t* __restrict__ ri,
                             const int32_t* __restrict__ Ts, const int32_t* __restrict__ Te,
                             const uint8_t* __restrict__ has, int32_t ncol, int64_t npanels,
                             int32_t panels_per_group, int32_t* __restrict__ nonfinite, double* __restrict__ partial) {
    constexpr int WL = SPLIT > 1 ? NT / SPLIT + NT / 2 : NT;   // (tiles a workgroup densifies: panels_body)
    constexpr int W = WL * 16, W1 = W + kPanPad, NP = NT * (NT + 1) / 2;
    constexpr int kBufDoubles = PR * W1;
    static_assert((kBufDoubles * 8) % 16 == 0, "a panel buffer is whole 16-byte units");
    __shared__ __attribute__((aligned(16))) double panel[2 * kBufDoubles];
    __shared__ int32_t sT[2 * W];     // one panel's cells {Ts, Te}
    __shared__ int32_t sSafe[2];      // an entry of the panel whose cells went into sT in this phase / the phase before
    __shared__ double sStray[64 + 16 * (W / 16)];   // where the lanes without an entry put what they hold (lane, and a round's share: 16 doubles apart)
    // SPLIT workgroups (32 tiles: 4) share a range of panels, each with a quarter of the tile rows; every one of them
    // densifies the panels for itself.  Workgroups are dealt to the 8 XCDs round-robin, so blocks b and b + 8 meet in the
    // same L2: in every run of 8 * SPLIT blocks, block 8 h + k is part h of the run's k-th range -- what one of them has
    // fetched the others find in their XCD's L2 (as crossprod_tall_kernel's SPLIT does).
    int group = blockIdx.x / SPLIT, part = blockIdx.x - group * SPLIT;
    if (SPLIT > 1) {
        const int run = blockIdx.x / (8 * SPLIT), in_run = blockIdx.x - run * (8 * SPLIT);
        if ((run + 1) * (8 * SPLIT) <= (int)gridDim.x) {
            group = run * 8 + (in_run & 7);
            part = in_run >> 3;
        }
    }
    const int64_t P0 = (int64_t)group * panels_per_group;
    const int64_t P1 = P0 + panels_per_group < npanels ? P0 + panels_per_group : npanels;
    double* mine = partial + (size_t)group * NP * 256;
    // (every wavefront meets the same barriers: the panels of a workgroup are the same for all of them)
    // (One instruction stream for all wavefronts, the wavefront's number in a register.  A stream per wavefront, with
    // every LDS offset a constant, saves 136 vector instructions per panel and was slower; so were two streams that
    // multiply and move in opposite order on the two wavefronts of a SIMD.)
    if (SPLIT > 1 && ((ncol + 15) >> 4) / 2 + 1 != NAX) return;   // (not the instantiation for this width: the launcher's mistake, nothing is touched)
    panels_body<NT, NW, PR, SPLIT, WIDE, MODE, NAX>((xp_lds_f64*)panel, (xp_lds_i32*)sT, (xp_lds_i32*)sSafe, (xp_lds_f64*)sStray, has, x,
                                               ri, Ts, Te, ncol, npanels, P0, P1, part, nonfinite, mine);
}

// out(c1, c2) = the workgroups' results for that element (both triangles).  One wavefront per element: lane l
// adds the results of workgroups l, l + 64, ... in that order, then the 64 lane sums meet in a fixed butterfly --
// the same association on every run.  (One thread per element walking all ~1000 workgroups took 0.3-0.5 ms,
// as much as the tall kernel itself on a 1e6 x 64 matrix.)
__global__ __launch_bounds__(256) void crossprod_tall_combine_kernel(const double* __restrict__ partial,
                                                                     int32_t ngroups, int32_t nt, int32_t ncol,
                                                                     const int32_t* __restrict__ nonfinite,
                                                                     double* __restrict__ out) {
#pragma clang fp contract(off)
    if (*nonfinite) return;
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= (int64_t)ncol * ncol) return;   // (whole wavefronts)
    const int c1 = (int)(k / ncol), c2 = (int)(k % ncol);
    const int a = c1 < c2 ? c1 : c2, b = c1 < c2 ? c2 : c1;   // a <= b: tile pair (a / 16, b / 16)
    const int I = a >> 4, J = b >> 4;
    const int q = I * nt - I * (I - 1) / 2 + (J - I);
    const int np = nt * (nt + 1) / 2;
    const double* t = partial + (size_t)q * 256 + (size_t)(a & 15) * 16 + (b & 15);
    double sum = 0.0;
    for (int g = lane; g < ngroups; g += 64) sum += t[(size_t)g * np * 256];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) out[k] = sum;
}

// The same for the panel-table kernel's results (at most 512 workgroups): a workgroup of 1024 threads per ELEMS
// elements of a tile pair, thread = (slice of the workgroups' results, element): neighbouring threads read neighbouring
// doubles -- crossprod_tall_combine_kernel's lanes stride from one workgroup's result to the next, 68 us for 71 MB
// where this one takes a third.  Slice k adds the results of workgroups k, k + K, ... in that order, then the K sums
// are added in the order of k: the same association on every run.  Only elements on or above the diagonal are used,
// the others are their mirror images.
template <int ELEMS>   // 256, 128 or 64: fewer elements per workgroup where there are few tile pairs
__global__ __launch_bounds__(1024) void crossprod_panels_combine_kernel(const double* __restrict__ partial, int32_t ngroups,
                                                                        int32_t nt, int32_t ncol,
                                                                        const int32_t* __restrict__ nonfinite,
                                                                        double* __restrict__ out) {
#pragma clang fp contract(off)
    constexpr int K = 1024 / ELEMS, PER = 256 / ELEMS;   // slices; workgroups per tile pair
    __shared__ double part[K][ELEMS];
    if (*nonfinite) return;
    const int q = blockIdx.x / PER, np = nt * (nt + 1) / 2;
    const int k = threadIdx.x / ELEMS, e = (blockIdx.x % PER) * ELEMS + threadIdx.x % ELEMS;
    const double* t = partial + (size_t)q * 256 + e;
    double sum = 0.0;
    for (int g = k; g < ngroups; g += K) sum += t[(size_t)g * np * 256];
    part[k][threadIdx.x % ELEMS] = sum;
    __syncthreads();
    if (k != 0) return;
    double total = part[0][threadIdx.x];
#pragma unroll
    for (int j = 1; j < K; ++j) total += part[j][threadIdx.x];
    int I = 0, rem = q;   // tile pair (I <= J) number q, row by row
    while (rem >= nt - I) {
        rem -= nt - I;
        ++I;
    }
    const int J = I + rem;
    const int a = 16 * I + (e >> 4), b = 16 * J + (e & 15);
    if (a <= b && b < ncol) {
        out[(size_t)a * ncol + b] = total;
        out[(size_t)b * ncol + a] = total;
    }
}

// nsplit slices of `width` result rows each: width <= kXMaxWidth, every slice non-empty.
// More than one slice when the LDS accumulators of a whole result column do not fit, and when
// there are too few columns to fill the chip: then a column's work is shared by nsplit waves, as
// long as the slices of a row keep about 32 entries on average (below that the lanes of a wave
// run empty and nothing is gained).
void crossprod_split(int32_t nrow, int32_t ncol, int64_t nnz, int32_t* nsplit, int32_t* width) {
    int ns = (ncol + kXMaxWidth - 1) / kXMaxWidth;
    if (ncol < 768 && ns == 1) {   // (workgroups of 4 wavefronts: 768 of them are half a round on 256 CUs)
        const int64_t avg_row = nrow > 0 ? nnz / nrow : 0;
        int64_t want = (768 + ncol - 1) / ncol;
        if (want > avg_row / 32) want = avg_row / 32;
        if (want > 8) want = 8;
        if (want > ncol) want = ncol;
        if (want > 1) ns = (int)want;
    }
    const int w = (ncol + ns - 1) / ns;
    *width = w;
    *nsplit = (ncol + w - 1) / w;                         // drops slices that would start beyond ncol
}

static inline size_t xp_align(size_t v) { return (v + 255) / 256 * 256; }

static int tall_tiles(int32_t ncol) {   // column tiles the kernels are instantiated for: 1, 2, 3, 4, 6, 8, 12, 16, 24 or 32
    const int nt = (ncol + 15) / 16;
    return nt <= 4 ? (nt < 1 ? 1 : nt) : (nt <= 6 ? 6 : (nt <= 8 ? 8 : (nt <= 12 ? 12 : (nt <= 16 ? 16 : (nt <= 24 ? 24 : 32)))));
}

// Does the matrix-core form pay?  Its time does not depend on how sparse the matrix is -- every 64-row panel is a dense
// rank-64 update of ncol x ncol (padded to whole tiles) -- while the exact forms pay per PRODUCT and per entry of a
// column's serial walk.  A model in milliseconds, calibrated on round 4's edge sweep (1e6 rows, columns of exactly 4096
// entries: 256 columns tall 1.76 against exact 0.45 ms; 64 columns 0.23 against 0.41; profiles/r04_form_edges.json):
//   exact: 0.04 + 1e-4 per entry of a column (the walk) + products / 4e8 per ms, products ~ nnz^2 / nrow for rows alike
//   tall : 0.13 + nrow * width^2 / 3.8e10 per ms (the panels' multiply-adds) + 12 B per entry at 3.9 TB/s
// Rounds 2-3 asked for columns of >= 4096 entries only, which sent sparse wide matrices (256 columns, 0.4 % dense) to a
// form four times slower.
// Which tile counts take the panel-table kernel: 8, 12 and 16 (97-256 columns; at 4 and 6 tiles it was measured and does
// not pay: 1e6 x 64 / 96 0.276 / 0.497 ms against 0.272 / 0.417 of the kernel that walks the CSC arrays).  (RSP_CROSSPROD_PANEL_TABLE=0: none --
// round 3's kernel everywhere, for comparisons; =1: 16 tiles only)
static bool panel_table_tiles(int ntiles) {
    const char* pm = getenv("RSP_CROSSPROD_PANEL_TABLE");
    const int level = pm ? atoi(pm) : 2;
    return level >= 2 ? (ntiles == 8 || ntiles == 12 || ntiles == 16 || ntiles == 24 || ntiles == 32) : (level == 1 && ntiles == 16);
}

static bool tall_pays(int32_t nrow, int32_t ncol, int64_t nnz) {
    const char* always = getenv("RSP_CROSSPROD_TALL_ALWAYS");   // (edge measurements: the round 2-3 rule, columns of >= 4096 entries)
    if (always && always[0] == '1') return true;
    const double len = (double)nnz / (double)ncol;
    const double rows = nrow > 0 ? (double)nrow : 1.0;
    const double products = (double)nnz * (double)nnz / rows + (double)nnz;
    const double t_exact = 0.04 + len * 1.0e-4 + products / 4.0e8;
    const double width = 16.0 * tall_tiles(ncol);
    double t_tall = 0.13 + rows * width * width / 3.8e10 + 12.0 * (double)nnz / 3.9e9;
    // (8 / 12 / 16 tiles, the panel-table kernel.  1e6 rows x 256 at 0.4 / 10 / 50 / 90 % density 1.20 / 1.32 / 1.53 / 1.81
    // ms, x 192 at 10 / 50 / 90 % 0.85 / 1.01 / 1.29, x 128 0.48 / 0.57 / 0.77; 4e6 rows at 5 % 4.68 / 2.98 / 1.55 ms; 24 / 32
    // tiles (257-512 columns): 1e6 rows x 384 at 10 / 50 % 3.47 / 3.87 ms, x 512 6.2 ms at either --
    // profiles/r04_crossprod_panels.json, r04_form_edges_crossprod.json; round 5, local tile order: x 384 3.3 / 3.5 ms,
    // x 512 5.5 ms, profiles/r05_crossprod_wide.json)
    const int nt = tall_tiles(ncol);
    // (24 / 32 tiles: one instantiation per REAL tile count rt, rt / 2 + 1 pairs per tile row instead of nt / 2 + 1 -- the
    // multiply-adds are 64 % of the kernel's time at full width: 257 / 320 / 385 / 448 columns 0.80 / 0.87 / 0.85 / 0.90 of
    // the full width's time, profiles/r05_crossprod_widths.json)
    const int rt = (ncol + 15) / 16;
    const double partial = nt >= 24 ? 0.36 + 0.64 * (double)(rt / 2 + 1) / (double)(nt / 2 + 1) : 1.0;
    if (panel_table_tiles(nt))
        t_tall = 0.09 + partial * rows * width * width / (nt == 32 ? 4.8e10 : (nt == 24 ? 4.8e10 : (nt == 16 ? 6.0e10 : (nt == 12 ? 5.2e10 : 4.5e10)))) + (nt == 32 ? 0.0 : 2.7e-9 * (double)nnz);
    return t_tall <= t_exact;
}

hipError_t plan_crossprod(int32_t nrow, int32_t ncol, int64_t nnz, bool exact, CrossprodLayout* L) {
    memset(L, 0, sizeof(*L));
    crossprod_split(nrow, ncol > 0 ? ncol : 1, nnz, &L->nsplit, &L->width);
    L->tall = !exact && ncol >= 1 && ncol <= kTallMaxCols && nnz / ncol >= kTallMinColumnLength &&
              (tall_tiles(ncol) <= 16 || panel_table_tiles(tall_tiles(ncol))) && tall_pays(nrow, ncol, nnz);
    if (L->tall) {   // one row-major form, unsliced, shared with the exact kernel that stands by
        L->nsplit = 1;
        L->width = ncol;
        L->ntiles = tall_tiles(ncol);
        const int64_t npanels = ((int64_t)nrow + kTallRows - 1) / kTallRows;
        // one round of workgroups: what fits on the chip at this tile count (registers / LDS per workgroup)
        static const int per_cu[17] = {0, 5, 5, 5, 2, 0, 2, 0, 2, 0, 0, 0, 1, 0, 0, 0, 1};
        // (16 tiles: two workgroups share every row range, so half as many ranges make one round)
        const int cu_share = L->ntiles <= 16 ? per_cu[L->ntiles] : 1;
        int64_t max_groups = 256 * cu_share < kTallMaxGroups ? 256 * cu_share : kTallMaxGroups;
        if (L->ntiles == 16) max_groups /= kTallSplit16;
        if (L->ntiles == 12) max_groups /= kTallSplit12;
        int64_t per = (npanels + max_groups - 1) / max_groups;
        if (per < 1) per = 1;
        L->panels_per_group = (int32_t)per;
        L->ngroups = (int32_t)((npanels + per - 1) / per);
        if (L->ngroups < 1) L->ngroups = 1;
        // 16 column tiles: panels of 32 rows through a panel table (crossprod_panels_kernel), one workgroup per CU
        if (panel_table_tiles(L->ntiles)) {
            L->panel_table = true;
            L->panel_rows = L->ntiles > 16 ? 16 : kPanRows;   // (24 / 32 tiles: two buffers of 32 x 384 / 512 do not fit the LDS)
            L->npanels = ((int64_t)nrow + L->panel_rows - 1) / L->panel_rows;
            const char* gr = getenv("RSP_CROSSPROD_PANEL_GROUPS");
            // (8 tiles: two workgroups per CU; 24 / 32 tiles: three / four workgroups share a range of panels)
            int64_t groups = gr ? atoll(gr) : (L->ntiles == 8 ? 512 : (L->ntiles == 32 ? 64 : (L->ntiles == 24 ? 85 : 256)));
            if (groups < 1) groups = 1;
            if (groups > kTallMaxGroups) groups = kTallMaxGroups;
            int64_t pper = (L->npanels + groups - 1) / groups;
            if (pper < 1) pper = 1;
            L->panels_per_group = (int32_t)pper;
            L->ngroups = (int32_t)((L->npanels + pper - 1) / pper);
            if (L->ngroups < 1) L->ngroups = 1;
        }
    }
    const size_t nv1 = (size_t)nrow * (size_t)L->nsplit + 1;   // virtual rows + 1
    if (nv1 > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t temp = exclusive_scan_temp_bytes((int64_t)nv1);   // (scan.hip: hand-written since round 4)
    size_t off = 0;
    L->rp_off = off;     off = xp_align(off + nv1 * 4);
    L->cursor_off = off; off = xp_align(off + nv1 * 4);
    L->rc_off = off;     off = xp_align(off + (size_t)nnz * 4);
    L->rx_off = off;     off = xp_align(off + (size_t)nnz * 8);
    L->temp_off = off;   off = xp_align(off + temp);
    L->temp_bytes = temp;
    if (L->tall) {
        const size_t np = (size_t)L->ntiles * (L->ntiles + 1) / 2;
        L->partial_off = off; off = xp_align(off + (size_t)L->ngroups * np * 256 * 8);
        L->flag_off = off;    off = xp_align(off + 4);
        if (L->panel_table) {
            L->table_off = off; off = xp_align(off + 2 * (size_t)L->npanels * (size_t)ncol * 4);   // Ts, then Te
            L->has_off = off;   off = xp_align(off + (size_t)L->npanels + 4);
        }
    }
    L->total_bytes = off;
    return hipSuccess;
}

template <int NT, int NW, int SPLIT = 1>
static void launch_tall(const CrossprodLayout& L, const double* d_x, const int32_t* d_i, const int32_t* d_p,
                        int32_t nrow, int32_t ncol, int64_t nnz, int32_t* flag, double* partial,
                        hipStream_t stream) {
    hipLaunchKernelGGL((crossprod_tall_kernel<NT, NW, SPLIT>), dim3((unsigned)L.ngroups * SPLIT), dim3(NW * 64), 0, stream,
                       d_x, d_i, d_p, nrow, ncol, nnz, L.panels_per_group, flag, partial);
}

// part: kXpAll -- everything, nothing synchronises (the exact kernels stand by on the tall form's flag); kXpTallOnly --
// the tall form alone (a caller that synchronises anyway looks at the flag itself: the eight launches of the exact
// kernels that stand by are 50 us, a quarter of a 1e6 x 32 call); kXpExactOnly -- the exact kernels alone,
// unconditionally.
hipError_t launch_crossprod_rows(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t nrow,
                                 int32_t ncol, int64_t nnz, double* d_out, const CrossprodLayout& L, void* ws,
                                 hipStream_t stream, int part) {
    if (ncol <= 0) return hipSuccess;
    int32_t* rp = (int32_t*)((char*)ws + L.rp_off);
    int32_t* cursor = (int32_t*)((char*)ws + L.cursor_off);
    int32_t* rc = (int32_t*)((char*)ws + L.rc_off);
    double* rx = (double*)((char*)ws + L.rx_off);
    const int nsplit = L.nsplit, width = L.width;
    const size_t nv1 = (size_t)nrow * (size_t)nsplit + 1;
    hipError_t e = hipSuccess;
    const int32_t* run_if = nullptr;
    if (L.tall && part != kXpExactOnly) {
        int32_t* flag = (int32_t*)((char*)ws + L.flag_off);
        double* partial = (double*)((char*)ws + L.partial_off);
        // ONE memset: the flag, and behind it in the workspace the panel table (0, 0: nothing of column c in panel P)
        // and which panels hold entries (round 5: three memsets before)
        const size_t zero_end = L.panel_table ? L.has_off + ((size_t)L.npanels + 3) / 4 * 4 : L.flag_off + 4;
        e = hipMemsetAsync(flag, 0, zero_end - L.flag_off, stream);
        if (e != hipSuccess) return e;
        if (L.panel_table) {
            // the panel table (and which panels hold entries), then the matrix-core kernel that reads x / i through it
            int32_t* Ts = (int32_t*)((char*)ws + L.table_off);
            int32_t* Te = Ts + (size_t)L.npanels * (size_t)ncol;
            uint8_t* has = (uint8_t*)((char*)ws + L.has_off);
            const int want_y = (ncol + 3) / 4;
            int xparts = (int)(nnz / ((int64_t)ncol * 2048));   // (a part walks ~2048 entries or more)
            if (xparts > 4096 / want_y) xparts = 4096 / want_y;   // (1024 ... 16384 blocks: 0.22 ... 0.16 ms, flat from 1792 on)
            if (xparts < 1) xparts = 1;
            if (nnz > 0)
                hipLaunchKernelGGL(xp_panel_table_kernel, dim3((unsigned)xparts, (unsigned)want_y), dim3(256), 0, stream,
                                   d_i, d_p, L.panel_rows == 16 ? 4 : 5, ncol, nnz, L.npanels, Ts, Te);
            hipLaunchKernelGGL(xp_panel_has_kernel, dim3((unsigned)((L.npanels + 255) / 256), (unsigned)((ncol + 31) / 32)),
                               dim3(256), 0, stream, (const int32_t*)Te, ncol, L.npanels, L.ntiles,
                               L.ntiles == 32 ? 4 : (L.ntiles == 24 ? 3 : 1), (uint32_t*)has);   // (the SPLIT of RSP_XP_LAUNCH below)
#define RSP_XP_LAUNCH_N(NT, NW, PR, SPLIT, WIDE, M, NAX)                                                                \
    hipLaunchKernelGGL((crossprod_panels_kernel<NT, NW, PR, SPLIT, WIDE, M, NAX>), dim3((unsigned)L.ngroups * SPLIT),   \
                       dim3(NW * 64), 0, stream, d_x, d_i, (const int32_t*)Ts, (const int32_t*)Te, (const uint8_t*)has, \
                       ncol, L.npanels, L.panels_per_group, flag, partial)
#define RSP_XP_LAUNCH(NT, NW, PR, SPLIT, WIDE, M) RSP_XP_LAUNCH_N(NT, NW, PR, SPLIT, WIDE, M, NT / 2 + 1)
// 24 / 32 tiles: the instantiation for the matrix's REAL tile count RT (NAX = RT / 2 + 1 pairs per tile row; panels_body)
#define RSP_XP_LAUNCH_WIDE(NT, SPLIT, WIDE)                                                                            \
    switch (((ncol + 15) / 16) / 2 + 1) {                                                                              \
        case NT / 2 - 3: RSP_XP_LAUNCH_N(NT, 8, 16, SPLIT, WIDE, 0, NT / 2 - 3); break;                                 \
        case NT / 2 - 2: RSP_XP_LAUNCH_N(NT, 8, 16, SPLIT, WIDE, 0, NT / 2 - 2); break;                                 \
        case NT / 2 - 1: RSP_XP_LAUNCH_N(NT, 8, 16, SPLIT, WIDE, 0, NT / 2 - 1); break;                                 \
        case NT / 2: RSP_XP_LAUNCH_N(NT, 8, 16, SPLIT, WIDE, 0, NT / 2); break;                                         \
        case NT / 2 + 1: RSP_XP_LAUNCH_N(NT, 8, 16, SPLIT, WIDE, 0, NT / 2 + 1); break;                                 \
        /* a tile count this layout was not planned for (a change to tall_tiles without one here): REFUSE.  The kernel  \
           returns without writing its partial sums when NAX does not fit the width, and the reduce behind it would     \
           then hand out whatever the workspace held -- with RSP_OK */                                  \
        default: return hipErrorInvalidValue;                                                                            \
    }
            const bool wide = nnz >= (1ll << 29);   // (byte offsets of x beyond 32 bits)
            int mode = 0;
#ifdef RSP_XP_MODES   // (a measurement build: RSP_XP_PANELS_MODE = 1 no MFMAs, 2 no entries moved, 3 neither; 16 tiles only)
            if (const char* md = getenv("RSP_XP_PANELS_MODE")) mode = atoi(md);
            if (wide || L.ntiles != 16) mode = 0;
            if (mode == 1) RSP_XP_LAUNCH(16, 8, 32, 1, false, 1);
            if (mode == 2) RSP_XP_LAUNCH(16, 8, 32, 1, false, 2);
            if (mode == 3) RSP_XP_LAUNCH(16, 8, 32, 1, false, 3);
#endif
            if (mode == 0) {
                if (L.ntiles == 32) {          // (25 .. 32 real tiles: 13 .. 17 pairs per tile row)
                    if (wide) { RSP_XP_LAUNCH_WIDE(32, 4, true) } else { RSP_XP_LAUNCH_WIDE(32, 4, false) }
                } else if (L.ntiles == 24) {   // (17 .. 24 real tiles: 9 .. 13 pairs per tile row)
                    if (wide) { RSP_XP_LAUNCH_WIDE(24, 3, true) } else { RSP_XP_LAUNCH_WIDE(24, 3, false) }
                } else if (L.ntiles == 16) {
                    if (wide) RSP_XP_LAUNCH(16, 8, 32, 1, true, 0);
                    else RSP_XP_LAUNCH(16, 8, 32, 1, false, 0);
                } else if (L.ntiles == 12) {
                    if (wide) RSP_XP_LAUNCH(12, 12, 32, 1, true, 0);
                    else RSP_XP_LAUNCH(12, 12, 32, 1, false, 0);
                } else {
                    if (wide) RSP_XP_LAUNCH(8, 8, 32, 1, true, 0);
                    else RSP_XP_LAUNCH(8, 8, 32, 1, false, 0);
                }
            }
#undef RSP_XP_LAUNCH_WIDE
#undef RSP_XP_LAUNCH
#undef RSP_XP_LAUNCH_N
        } else
        switch (L.ntiles) {
            case 1: launch_tall<1, 4>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            case 2: launch_tall<2, 4>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            case 3: launch_tall<3, 4>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            case 4: launch_tall<4, 8>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            case 6: launch_tall<6, 8>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            case 8: launch_tall<8, 8>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            case 12: launch_tall<12, 16, kTallSplit12>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
            default: launch_tall<16, 16, kTallSplit16>(L, d_x, d_i, d_p, nrow, ncol, nnz, flag, partial, stream); break;
        }
        const int64_t outs = (int64_t)ncol * ncol;
        if (L.panel_table) {
            const unsigned pairs = (unsigned)(L.ntiles * (L.ntiles + 1) / 2);
            if (L.ntiles >= 16)
                hipLaunchKernelGGL(crossprod_panels_combine_kernel<128>, dim3(pairs * 2), dim3(1024), 0, stream, partial,
                                   L.ngroups, L.ntiles, ncol, flag, d_out);
            else if (L.ntiles == 12)
                hipLaunchKernelGGL(crossprod_panels_combine_kernel<128>, dim3(pairs * 2), dim3(1024), 0, stream, partial,
                                   L.ngroups, L.ntiles, ncol, flag, d_out);
            else
                hipLaunchKernelGGL(crossprod_panels_combine_kernel<64>, dim3(pairs * 4), dim3(1024), 0, stream, partial,
                                   L.ngroups, L.ntiles, ncol, flag, d_out);
        } else
            hipLaunchKernelGGL(crossprod_tall_combine_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, stream,
                               partial, L.ngroups, L.ntiles, ncol, flag, d_out);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        run_if = flag;   // everything below only works if x holds a non-finite value
        if (part == kXpTallOnly) return hipSuccess;
    }
    // Standing by (run_if): SIX launches that read the flag and return -- no memset, no scan, no copy happens unless
    // a sum was not finite (round 5; before: eight operations, of which the memset of the cursors, the three scan
    // kernels and a device-to-device copy of the row offsets did their work in every call: 1e6 rows x 32 columns
    // 0.125 -> see profiles/r05_crossprod_standby.json).
    if (run_if)
        hipLaunchKernelGGL(xp_zero_if_kernel, dim3((unsigned)std::min<size_t>((nv1 + 1023) / 1024, 4096)), dim3(256), 0,
                           stream, cursor, (int64_t)nv1, run_if);
    else
        e = hipMemsetAsync(cursor, 0, nv1 * 4, stream);
    if (e != hipSuccess) return e;
    const int want = (ncol + 3) / 4;
    // ~4096 wavefronts: columns over y (4 per block), parts of a column over x
    const int ycols = want < 1024 ? want : 1024;
    int xparts = (int)(nnz / ((int64_t)(ncol > 0 ? ncol : 1) * 2048));   // a part walks ~2048 entries or more
    if (xparts > 1024 / ycols) xparts = 1024 / ycols;
    if (xparts < 1) xparts = 1;
    const dim3 cgrid((unsigned)xparts, (unsigned)ycols);
    if (nnz > 0)
        hipLaunchKernelGGL(xp_count_rows_kernel, cgrid, dim3(256), 0, stream, d_i, d_p, nrow, ncol, nsplit, width,
                           cursor, run_if);
    // (row offsets into rp AND, as the fill pass's cursors, over the counts themselves: the scan reads a tile whole
    // before it writes any of it)
    e = launch_exclusive_scan_i32(cursor, rp, (int64_t)nv1, 0, (char*)ws + L.temp_off, L.temp_bytes, stream, run_if,
                                  cursor);
    if (e != hipSuccess) return e;
    if (nnz > 0)
        hipLaunchKernelGGL(xp_fill_rows_kernel, cgrid, dim3(256), 0, stream, d_x, d_i, d_p, nrow, ncol, nsplit, width,
                           cursor, rc, rx, run_if);
    const long long grid = (long long)ncol * nsplit;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t rows_lds = ((size_t)width + 64) * 8 + (size_t)2 * kXRound * 64 * 12;
    static DynamicLdsLimit rows_limit;
    e = rows_limit.ensure((const void*)crossprod_rows_kernel,
                          (int)(((size_t)kXMaxWidth + 64) * 8 + (size_t)2 * kXRound * 64 * 12));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(crossprod_rows_kernel, dim3((unsigned)grid), dim3(256), rows_lds, stream,
                       d_x, d_i, d_p, rp, rc, rx, nrow, ncol, nsplit, width, d_out, run_if);
    return hipGetLastError();
}

hipError_t launch_crossprod(const double* d_x, const int32_t* d_i, const int32_t* d_p, int32_t ncol,
                            double* d_out, hipStream_t stream) {
    if (ncol <= 0) return hipSuccess;
    const int ntiles = (ncol + kXT - 1) / kXT;
    const long long pairs = (long long)ntiles * (ntiles + 1) / 2;
    if (pairs > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(crossprod_tiles_kernel, dim3((unsigned)pairs), dim3(256), 0, stream, d_x, d_i, d_p, ncol,
                       ntiles, d_out);
    return hipGetLastError();
}

}  // namespace rsp

// crossprod.hip -- Matrix::crossprod() on the device ("next" row f3 of SURVEY.md 8f).
//
// Reference inst/include/RcppSparse.h:159-194: dense ncol x ncol t(A) %*% A; every column
// pair is a sparse dot product by a sorted merge of the two row lists, res(c1,c2) +=
// x1 * x2 over the common rows in ascending row order (OpenMP over c1; the package's only
// parallel routine).  O(ncol^2) output: meant for matrices with few columns.
//
// Here one workgroup owns a 64 x 64 tile of the result (upper-triangular tile pairs only;
// the mirror image is written at the end).  It walks the rows of A in blocks of 64: the
// 64 + 64 columns of the tile pair each keep a cursor into their (ascending) row lists,
// the entries that fall into the current row block are scattered into two dense LDS
// panels [row][column] with per-row presence masks, and every thread accumulates its
// 4 x 4 sub-tile in registers over the rows that are present on both sides.  Row blocks
// in which none of the 128 columns has an entry are skipped (the next block starts at the
// smallest pending row).  Products are accumulated in ascending row order with a separate
// multiply and add (no FMA contraction), i.e. in the reference's order: results are
// bit-identical to the reference loop for finite data.  Only entries that are stored take
// part (presence masks), so a non-finite value never meets a structural zero.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

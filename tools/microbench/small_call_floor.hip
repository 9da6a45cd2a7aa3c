// What does a C2-sized call (80 MB of x, 8 MB of sums) cost at best, issued back to back on ONE stream?
// Pure streaming kernels with the production access pattern (one wavefront per chunk of `rows` 1 KiB rows,
// 16-byte nt buffer loads, 4 in flight), no column work at all:
//   read      reads x, stores nothing
//   readwrite reads x and stores 8 MB (every wave its share of 1e6 doubles, 512-byte store instructions)
//   empty     the same grid doing nothing (launch + drain floor)
// x rotates over 6 copies (480 MB > the 256 MB Infinity Cache), like bench.py does for C2.
// Reported: microseconds per launch for K launches back to back (HIP events around the K launches).
//   hipcc --offload-arch=gfx950 -O3 small_call_floor.hip -o small_call_floor && ./small_call_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>   // 0 read, 1 read + write, 2 empty
__global__ __launch_bounds__(256) void stream_kernel(const double* x, long nrows, int rows, double* out, long nout, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (MODE == 2 || w >= nwaves) return;
    const long r0 = w * rows;
    if (r0 >= nrows) return;
    const long r1 = (r0 + rows < nrows) ? r0 + rows : nrows;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + r0 * 128), 0, (int)((r1 - r0) * 1024), 0x00020000);
    d2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16, r * 1024, 2));
    double a0 = 0, a1 = 0;
    const int n = (int)(r1 - r0);
    for (int b = 0; b < n; b += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a0 += v[r].x; a1 += v[r].y;
            v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16, (b + r + 4) * 1024, 2));
        }
    }
    if (MODE == 1) {
        const long per = (nout + nwaves - 1) / nwaves;       // this wave's share of the output
        for (long c = w * per + lane; c < (w + 1) * per && c < nout; c += 64) out[c] = a0 + a1 + (double)c;
    } else if (a0 + a1 == 123.456) {
        out[w] = a0 + a1;
    }
}

int main(int argc, char** argv) {
    const long nnz = 10000000, ncol = 1000000;
    const int ncopies = 6, K = 400;
    const long nrows = (nnz + 127) / 128;
    double* x; double* out;
    CK(hipMalloc((void**)&x, (size_t)ncopies * nrows * 1024));
    CK(hipMalloc((void**)&out, ncol * 8));
    CK(hipMemset(x, 0, (size_t)ncopies * nrows * 1024));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rows : {8, 12, 16, 20, 32}) {
        const long nwaves = (nrows + rows - 1) / rows;
        const dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&](int k) {
                const double* xk = x + (size_t)(k % ncopies) * nrows * 128;
                if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, grid, block, 0, s, xk, nrows, rows, out, ncol, nwaves);
                if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, grid, block, 0, s, xk, nrows, rows, out, ncol, nwaves);
                if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, grid, block, 0, s, xk, nrows, rows, out, ncol, nwaves);
            };
            for (int k = 0; k < 20; ++k) launch(k);
            CK(hipStreamSynchronize(s));
            std::vector<float> t;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(a, s));
                for (int k = 0; k < K; ++k) launch(k);
                CK(hipEventRecord(b, s));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                t.push_back(ms * 1000.f / K);
            }
            std::sort(t.begin(), t.end());
            const char* name = mode == 0 ? "read" : mode == 1 ? "readwrite" : "empty";
            const double bytes = mode == 0 ? 8.0 * nnz : mode == 1 ? 8.0 * nnz + 8.0 * ncol : 0.0;
            printf("{\"rows_per_wave\": %d, \"waves\": %ld, \"kernel\": \"%s\", \"us_per_launch_back_to_back\": %.2f, \"GBps\": %.0f}\n",
                   rows, nwaves, name, t[t.size() / 2], bytes / t[t.size() / 2] / 1e3);
        }
    }
    return 0;
}

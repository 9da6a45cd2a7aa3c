// Does the read ceiling depend on how many wavefronts stream concurrently?  Same chunked read as
// read_ceiling.hip (nt loads), occupancy limited by a dynamic LDS allocation, deeper pipelines
// to keep the bytes in flight up.
//   hipcc --offload-arch=gfx950 -O3 read_occupancy.hip -o read_occupancy && ./read_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int DEPTH, int ROWB>   // ROWB: bytes per wave-row step = 1024 (one dwordx4 per lane) or 2048 (two)
__global__ __launch_bounds__(256) void chunked(const double* x, long nrows, int chunk_rows, double* out) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long r0 = w * chunk_rows;
    if (r0 >= nrows) return;
    const long r1 = (r0 + chunk_rows < nrows) ? r0 + chunk_rows : nrows;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + r0 * 128), 0, (int)((r1 - r0) * 1024), 0x00020000);
    d2 v[DEPTH];
    constexpr int PER = ROWB / 1024;
#pragma unroll
    for (int r = 0; r < DEPTH; ++r) {
        const int step = r / PER, sub = r % PER;
        v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16 * PER + sub * 16, step * ROWB, 2));
    }
    double a0 = 0, a1 = 0;
    const int n = (int)(r1 - r0);
    for (int b = 0; b < n; b += DEPTH) {
#pragma unroll
        for (int r = 0; r < DEPTH; ++r) {
            a0 += v[r].x;
            a1 += v[r].y;
            const int step = (b + r + DEPTH) / PER, sub = r % PER;
            v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16 * PER + sub * 16, step * ROWB, 2));
        }
    }
    if (a0 + a1 == 123.456) out[w] = a0 + a1 + pad[0];
}

template <class F>
double time_ms(F f, int reps = 11) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const long n = 1000000000L;
    const long nrows = (n + 127) / 128;
    double *x, *out;
    CK(hipMalloc(&x, nrows * 1024)); CK(hipMalloc(&out, 1 << 24));
    CK(hipMemset(x, 0, nrows * 1024));
    const double GB = n * 8.0 / 1e9;
    auto run = [&](const char* name, auto kern, int cr, int lds_kb) {
        const int blocks = (int)((nrows + (long)cr * 4 - 1) / ((long)cr * 4));
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), (size_t)lds_kb * 1024, 0, x, nrows, cr, out); });
        printf("%-40s chunk_rows %5d  LDS %3d KB/WG (%2d waves/CU)  %8.4f ms  %7.1f GB/s\n", name, cr, lds_kb,
               lds_kb ? (160 / lds_kb) * 4 : 32, ms, GB / ms * 1e3);
    };
    for (int cr : {256, 1024}) {
        for (int kb : {0, 26, 40, 53, 80}) {   // ~32, 24, 16, 12, 8 waves per CU
            run("depth 8, 1 KB steps", chunked<8, 1024>, cr, kb);
            run("depth 16, 1 KB steps", chunked<16, 1024>, cr, kb);
            run("depth 32, 1 KB steps", chunked<32, 1024>, cr, kb);
            run("depth 16, 2 KB steps", chunked<16, 2048>, cr, kb);
        }
    }
    return 0;
}

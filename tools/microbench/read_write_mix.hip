// What does the output side of a column-sum cost?  The streaming read of read_ceiling.hip
// (per-wave contiguous chunks, 8 rows in flight, nt loads) plus result stores in the shapes the
// production kernel produces or could produce:
//   mode 0  no stores (read ceiling)
//   mode 1  one store instruction of 64 consecutive doubles (512 B), 512-B aligned
//   mode 2  the same, starting 3 doubles past the alignment (what the dense path does today)
//   mode 3  8 consecutive doubles (64 B) per store instruction, aligned
//   mode 4  8 consecutive doubles per store instruction, unaligned (+3)
//   mode 5  one double per store instruction (single lane), consecutive over time
// `per` = rows read per store instruction; out doubles per row = width / per.
//   hipcc --offload-arch=gfx950 -O3 read_write_mix.hip -o read_write_mix && ./read_write_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void mix(const double* x, long nrows, int chunk_rows, int per, double* out,
                                           long out_per_chunk) {
    constexpr int DEPTH = 8;
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long r0 = w * chunk_rows;
    if (r0 >= nrows) return;
    const long r1 = (r0 + chunk_rows < nrows) ? r0 + chunk_rows : nrows;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + r0 * 128), 0, (int)((r1 - r0) * 1024), 0x00020000);
    d2 v[DEPTH];
#pragma unroll
    for (int r = 0; r < DEPTH; ++r)
        v[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16, r * 1024, 2));
    double a0 = 0, a1 = 0;
    const int n = (int)(r1 - r0);
    constexpr int WIDTH = (MODE == 1 || MODE == 2) ? 64 : ((MODE == 3 || MODE == 4) ? 8 : 1);
    constexpr int SKEW = (MODE == 2 || MODE == 4) ? 3 : 0;
    double* o = out + w * out_per_chunk + SKEW;
    int until = per;
    for (int b = 0; b < n; b += DEPTH) {
#pragma unroll
        for (int r = 0; r < DEPTH; ++r) {
            a0 += v[r].x;
            a1 += v[r].y;
            v[r] = __builtin_bit_cast(
                d2, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16, (b + r + DEPTH) * 1024, 2));
            if (MODE != 0 && --until == 0) {   // uniform
                until = per;
                if (lane < WIDTH) {
                    if (NT) __builtin_nontemporal_store(a0 + a1, o + lane);
                    else o[lane] = a0 + a1;
                }
                o += WIDTH;
            }
        }
    }
    if (a0 + a1 == 123.456) out[w] = a0 + a1;
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
    const int cr = 256;
    const int blocks = (int)((nrows + (long)cr * 4 - 1) / ((long)cr * 4));
    const long nchunks = (nrows + cr - 1) / cr;
    double *x, *out;
    const long out_cap = nchunks * (cr * 13L + 64) + 1024;     // up to 13 doubles per row
    CK(hipMalloc(&x, nrows * 1024)); CK(hipMalloc(&out, out_cap * 8));
    CK(hipMemset(x, 0, nrows * 1024)); CK(hipMemset(out, 0, out_cap * 8));
    auto run = [&](const char* name, auto kern, int per, int width) {
        const long opc = ((long)(cr / per + 1) * width + 63) / 64 * 64;   // 512-B aligned chunk regions
        const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, nrows, cr, per, out, opc); });
        const double outMB = (double)nchunks * (cr / per) * width * 8 / 1e6;
        printf("%-58s %8.4f ms   out %7.1f MB\n", name, ms, outMB);
    };
    run("read only", mix<0, false>, 1, 0);
    // ~12.8 result doubles per row (mean 10 nnz per column)
    run("64-wide aligned, every 5 rows", mix<1, false>, 5, 64);
    run("64-wide unaligned, every 5 rows", mix<2, false>, 5, 64);
    run("64-wide aligned nt, every 5 rows", mix<1, true>, 5, 64);
    run("64-wide unaligned nt, every 5 rows", mix<2, true>, 5, 64);
    // ~1.3 per row (mean 100)
    run("64-wide aligned, every 50 rows", mix<1, false>, 50, 64);
    run("64-wide unaligned, every 50 rows", mix<2, false>, 50, 64);
    run("64-wide aligned nt, every 50 rows", mix<1, true>, 50, 64);
    run("64-wide unaligned nt, every 50 rows", mix<2, true>, 50, 64);
    run("8-wide aligned, every 6 rows", mix<3, false>, 6, 8);
    run("8-wide unaligned, every 6 rows", mix<4, false>, 6, 8);
    run("1-wide, every row", mix<5, false>, 1, 1);
    run("1-wide nt, every row", mix<5, true>, 1, 1);
    // ~0.13 per row (mean 1000)
    run("1-wide, every 8 rows", mix<5, false>, 8, 1);
    run("64-wide aligned, every 256 rows", mix<1, false>, 256, 64);
    run("64-wide aligned nt, every 256 rows", mix<1, true>, 256, 64);
    run("read only (again)", mix<0, false>, 1, 0);
    return 0;
}

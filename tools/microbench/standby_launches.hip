// What do launches that do nothing cost behind a kernel that did something?  (crossprod's exact kernels stand by
// behind the tall form's flag: launch_crossprod_rows, crossprod.hip.)  One "call" = a small real kernel (~20 us of
// streaming) followed by
//   none      nothing
//   chain8    eight kernels that read a flag and return (grids as in the stand-by chain: 1 ... 4096 blocks)
//   chain3    three of them
//   one       one of them
//   coop      ONE kernel launched with hipLaunchCooperativeKernel (1024 blocks of 256) that reads the flag and returns
//   memops    hipMemsetAsync(4 MB) + hipMemcpyAsync(4 MB, device to device): what the chain holds beside kernels
// Reported: microseconds per call, K calls back to back on one stream (HIP events around the K calls).
//   hipcc --offload-arch=gfx950 -O3 standby_launches.hip -o standby_launches && ./standby_launches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void work_kernel(const double* x, long n, double* out) {
    long k = (long)blockIdx.x * 256 + threadIdx.x;
    double a = 0;
    for (; k < n; k += (long)gridDim.x * 256) a += x[k];
    if (a == 123.456) out[0] = a;
}

__global__ __launch_bounds__(256) void standby_kernel(const int* run_if, double* out) {
    if (*run_if == 0) return;
    out[blockIdx.x] = 1.0;
}

int main() {
    const long n = 8 << 20;   // 64 MB
    const int K = 300;
    double* x; double* out; int* flag; char* a4; char* b4;
    CK(hipMalloc((void**)&x, n * 8)); CK(hipMemset(x, 0, n * 8));
    CK(hipMalloc((void**)&out, 1 << 20)); CK(hipMalloc((void**)&flag, 4)); CK(hipMemset(flag, 0, 4));
    CK(hipMalloc((void**)&a4, 4 << 20)); CK(hipMalloc((void**)&b4, 4 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grids[8] = {1, 128, 977, 1, 977, 128, 128, 4096};
    const char* names[] = {"none", "chain8", "chain3", "one", "coop", "memops"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) CK(hipEventRecord(e0, s));
            for (int k = 0; k < (pass ? K : 20); ++k) {
                hipLaunchKernelGGL(work_kernel, dim3(2048), dim3(256), 0, s, x, n, out);
                const int nl = mode == 1 ? 8 : mode == 2 ? 3 : mode == 3 ? 1 : 0;
                for (int j = 0; j < nl; ++j)
                    hipLaunchKernelGGL(standby_kernel, dim3(grids[j]), dim3(256), 0, s, (const int*)flag, out);
                if (mode == 4) {
                    const int* f = flag; double* o = out;
                    void* args[] = {(void*)&f, (void*)&o};
                    CK(hipLaunchCooperativeKernel((const void*)standby_kernel, dim3(1024), dim3(256), args, 0, s));
                }
                if (mode == 5) {
                    CK(hipMemsetAsync(a4, 0, 4 << 20, s));
                    CK(hipMemcpyAsync(b4, a4, 4 << 20, hipMemcpyDeviceToDevice, s));
                }
            }
            if (pass == 1) CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-7s %8.2f us per call\n", names[mode], 1000.0 * ms / K);
    }
    return 0;
}

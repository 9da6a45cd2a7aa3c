// How much other work issues in the shadow of v_mfma_f64_16x16x4_f64 on gfx950?  Every wavefront runs a loop of 17
// independent MFMAs per step with K other instructions after each MFMA (K = 0, 1, 2, 4, 8; kinds: 32-bit VALU, LDS
// write, LDS read); 2 wavefronts per SIMD (512 threads per workgroup, one workgroup per CU), as crossprod_panels_kernel
// runs.  Prints microseconds per configuration; the K = 0 line is the matrix-core time.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int KIND, int K>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void shadow_kernel(double* out, int steps) {
    __shared__ double lds[8192];
    const int tid = threadIdx.x;
    v4f64 acc[17];
#pragma unroll
    for (int s = 0; s < 17; ++s) acc[s] = v4f64{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + tid * 1e-9, b = 1.0 - tid * 1e-9;
    int v[8] = {tid, tid + 1, tid + 2, tid + 3, tid + 4, tid + 5, tid + 6, tid + 7};
    double dv = 0.0;
    lds[tid] = tid;
    lds[tid + 512] = 0;
    __syncthreads();
    for (int it = 0; it < steps; ++it) {
#pragma unroll
        for (int s = 0; s < 17; ++s) {
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[s], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(tid));
                if (KIND == 1) asm volatile("ds_write_b64 %0, %1" ::"v"((tid & 63) * 8 + 4096 * (s & 1)), "v"(a) : "memory");
                if (KIND == 2) { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((tid * 8) & 4095) : "memory"); dv += 0 * t; }
                if (KIND == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(*(long long*)&v[(k & 3) * 2]) : "v"(tid) : "vcc");
                if (KIND == 4) asm volatile("s_nop 0");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double sum = dv;
#pragma unroll
    for (int s = 0; s < 17; ++s) sum += acc[s][0] + acc[s][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
    if (sum == 12345.678) out[0] = sum + lds[tid];
}

template <int KIND, int K>
static void run(const char* kind, double* d_out) {
    const int steps = 2000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    shadow_kernel<KIND, K><<<256, 512>>>(d_out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    shadow_kernel<KIND, K><<<256, 512>>>(d_out, steps);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double mfmas_per_simd = 2.0 * steps * 17;   // two wavefronts per SIMD
    printf("{\"kind\": \"%s\", \"per_mfma\": %d, \"us\": %.1f, \"cycles_per_mfma_at_2.4GHz\": %.1f}\n", kind, K, ms * 1e3,
           ms * 1e-3 * 2.4e9 / mfmas_per_simd);
}

int main() {
    double* d_out;
    (void)hipMalloc(&d_out, 8);
    run<0, 0>("none", d_out);
    run<0, 1>("valu32", d_out); run<0, 2>("valu32", d_out); run<0, 4>("valu32", d_out); run<0, 8>("valu32", d_out); run<0, 12>("valu32", d_out);
    run<3, 1>("valu_mad64", d_out); run<3, 2>("valu_mad64", d_out); run<3, 4>("valu_mad64", d_out);
    run<1, 1>("lds_write", d_out); run<1, 2>("lds_write", d_out); run<1, 4>("lds_write", d_out);
    run<2, 1>("lds_read", d_out); run<2, 2>("lds_read", d_out); run<2, 4>("lds_read", d_out);
    run<4, 4>("s_nop", d_out); run<4, 12>("s_nop", d_out);
    return 0;
}

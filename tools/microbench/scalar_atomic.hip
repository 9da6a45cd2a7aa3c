// Does gfx950 execute scalar-cache atomics (s_atomic_add ... glc)?  One ticket per wave through the
// scalar unit; checks that the tickets are a permutation of 0..nwaves-1 and times the hand-out.
// Build: hipcc --offload-arch=gfx950 -O2 -o scalar_atomic scalar_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void take_tickets(unsigned* counter, unsigned* got, int per_wave) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int k = 0; k < per_wave; ++k) {
        unsigned t = 1;   // data in, pre-op value out
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(counter) : "memory");
        if ((threadIdx.x & 63) == 0) got[t] = (unsigned)wave + 1;
    }
}

int main() {
    const int blocks = 2048, threads = 256, per_wave = 16;
    const int nwaves = blocks * threads / 64, total = nwaves * per_wave;
    unsigned *counter, *got;
    hipMalloc(&counter, 4);
    hipMalloc(&got, (size_t)(total + 64) * 4);
    hipMemset(counter, 0, 4);
    hipMemset(got, 0, (size_t)(total + 64) * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(take_tickets, dim3(blocks), dim3(threads), 0, 0, counter, got, per_wave);
    hipEventRecord(b);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(e)); return 1; }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned final_count = 0;
    hipMemcpy(&final_count, counter, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> h(total);
    hipMemcpy(h.data(), got, (size_t)total * 4, hipMemcpyDeviceToHost);
    const bool all = std::all_of(h.begin(), h.end(), [](unsigned v) { return v != 0; });
    printf("scalar atomics: counter = %u (want %d), every ticket handed out exactly once: %s, %.1f ns per ticket\n",
           final_count, total, all ? "yes" : "NO", ms * 1e6 / total);
    return (final_count == (unsigned)total && all) ? 0 : 2;
}

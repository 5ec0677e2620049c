// tools/ubench/conc.hip -- how many kernels of ONE process run side by side on gfx950?  K streams, one small long-running kernel on each (64 single-wavefront
// workgroups spinning for ~1 ms: the chip is nowhere near full), wall time of the K launches together.  If all K run concurrently the time stays ~1 ms; if the
// hardware (or the runtime's queue mapping) serialises beyond C of them it grows as ceil(K / C) ms.  Run with GPU_MAX_HW_QUEUES = 4 (HIP's default), 8, 16.
// build: hipcc --offload-arch=gfx950 -O3 -o conc tools/ubench/conc.hip ; run: GPU_MAX_HW_QUEUES=8 ./conc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(uint64_t ticks, uint64_t *out) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, constant
    uint64_t t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(32); t = __builtin_amdgcn_s_memrealtime(); }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}

int main() {
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s: K kernels on K streams, each 64 workgroups x 64 threads spinning 1.0 ms; wall ms for all K (best of 5)\n", q ? q : "(unset)");
    uint64_t *d_out;
    hipMalloc(&d_out, 8 * 64);
    std::vector<hipStream_t> st(32);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int K : {1, 2, 3, 4, 5, 6, 8, 12, 16, 32}) {
        double best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, st[k], (uint64_t)100000, d_out + k);
            hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep && ms < best) best = ms;
        }
        printf("  K = %2d   %6.2f ms   -> about %.1f side by side\n", K, best, K / best);
    }
    return 0;
}

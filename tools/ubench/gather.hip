// tools/ubench/gather.hip -- scattered 2-byte loads from an L2-resident table (the connection-cost gather of
// k_tokenize_pool): cache-line requests per cycle per CU as a function of wavefronts per CU and loads in flight per lane.
// build: hipcc --offload-arch=gfx950 -O3 -o gather tools/ubench/gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdint>
#include <cstdio>
#include <vector>

// GROUP: that many neighbouring lanes read (different 2-byte elements of) the same 128-byte line
template <int INFLIGHT, int GROUP = 1>
__global__ void k_gather(const int16_t *tab, uint32_t mask, int rounds, uint64_t *cycles, int *sink) {
    uint32_t x = ((blockIdx.x * blockDim.x + threadIdx.x) / GROUP) * 2654435761u + 12345u;
    const uint32_t within = (threadIdx.x % GROUP) * (64 / GROUP);
    int acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        int16_t v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) { x = x * 1664525u + 1013904223u; v[k] = tab[GROUP == 1 ? (x >> 8) & mask : ((((x >> 8) & mask) & ~63u) | within)]; }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) acc += v[k];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&cycles[blockIdx.x], (unsigned long long)(t1 - t0));
    if (acc == 0x12345678) *sink = acc;
}

int main() {
    const size_t n = 1u << 21;  // 4 MB of int16 (the connection matrix is 3.46 MB)
    int16_t *tab; uint64_t *cyc; int *sink;
    hipMalloc(&tab, n * 2); hipMemset(tab, 1, n * 2); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
    const int rounds = 256;
    for (size_t tab_n : {(size_t)1 << 15, (size_t)1 << 17, (size_t)1 << 19, (size_t)1 << 20, (size_t)1 << 21})
    for (int waves : {16}) {
        for (int inflight : {4}) {
            hipMemset(cyc, 0, 256 * 8);
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(cyc, 0, 256 * 8);
                if (inflight == 1) hipLaunchKernelGGL(k_gather<1>, dim3(256), dim3(64 * waves), 0, 0, tab, (uint32_t)(tab_n - 1), rounds * 8, cyc, sink);
                if (inflight == 4) hipLaunchKernelGGL(k_gather<4>, dim3(256), dim3(64 * waves), 0, 0, tab, (uint32_t)(tab_n - 1), rounds * 2, cyc, sink);
                if (inflight == 8) hipLaunchKernelGGL(k_gather<8>, dim3(256), dim3(64 * waves), 0, 0, tab, (uint32_t)(tab_n - 1), rounds, cyc, sink);
                hipDeviceSynchronize();
            }
            std::vector<uint64_t> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : h) avg += c; avg /= 256;
            const double reqs = (double)waves * 64 * rounds * 8;  // lane-requests per workgroup (= per CU: 256 workgroups, one per CU)
            printf("table %zu KB: waves/CU %2d, loads in flight per lane %d: %.0f cycles, %.2f lane-requests per cycle per CU, %.0f cycles per wave-instruction\n", tab_n * 2 / 1024, waves, inflight, avg,
                   reqs / avg, avg / (rounds * 8.0 ));
        }
    }
    // lanes sharing lines: G neighbouring lanes per 128-byte line, 4 MB table, 16 waves per CU, 4 loads in flight
    auto run = [&](auto kern, int g) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(cyc, 0, 256 * 8);
            hipLaunchKernelGGL(kern, dim3(256), dim3(64 * 16), 0, 0, tab, (uint32_t)(n - 1), rounds * 2, cyc, sink);
            hipDeviceSynchronize();
        }
        std::vector<uint64_t> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= 256;
        printf("lanes per line %2d: %.0f cycles, %.2f lane-requests per cycle per CU, %.0f cycles per wave-instruction\n", g, avg, 16.0 * 64 * rounds * 8 / avg, avg / (rounds * 8.0));
    };
    run(k_gather<4, 1>, 1); run(k_gather<4, 2>, 2); run(k_gather<4, 4>, 4); run(k_gather<4, 8>, 8); run(k_gather<4, 16>, 16); run(k_gather<4, 64>, 64);
    return 0;
}

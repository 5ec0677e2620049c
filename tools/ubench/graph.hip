// tools/ubench/graph.hip -- is a hipGraph worth it for the per-batch chain (tokenize kernel -> scan -> compact, arguments that change with every call)?
// Three dependent ~10 us kernels per chain on one stream, 2000 chains: host time per chain and elapsed time per chain for
//   (a) three hipLaunchKernelGGL calls, (b) one hipGraphLaunch of the captured chain, (c) the same with hipGraphExecKernelNodeSetParams on all three
//   nodes before every launch (what changing pointers would need).
// build: hipcc --offload-arch=gfx950 -O3 -o graph tools/ubench/graph.hip ; run: ./graph
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_work(int *p, int spins) {
    int v = p[threadIdx.x];
    for (int i = 0; i < spins; ++i) v = v * 1664525 + 1013904223;
    p[threadIdx.x] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    int *buf; hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
    hipStream_t st; hipStreamCreate(&st);
    const int N = 2000, spins = (getenv("SPINS") ? atoi(getenv("SPINS")) : 400);
    auto chain = [&] { for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, buf, spins); };
    for (int i = 0; i < 50; ++i) chain();
    hipStreamSynchronize(st);
    double t0 = now_us(), host = 0;
    for (int i = 0; i < N; ++i) { const double a = now_us(); chain(); host += now_us() - a; }
    hipStreamSynchronize(st);
    const double direct = (now_us() - t0) / N;
    printf("three launches per chain:         host %.1f us per chain, elapsed %.1f us per chain\n", host / N, direct);

    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = now_us(); host = 0;
    for (int i = 0; i < N; ++i) { const double a = now_us(); hipGraphLaunch(ge, st); host += now_us() - a; }
    hipStreamSynchronize(st);
    printf("one graph launch per chain:       host %.1f us per chain, elapsed %.1f us per chain\n", host / N, (now_us() - t0) / N);

    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g, nodes.data(), &nn);
    int spins2 = spins; int *p2 = buf; void *args[2] = {&p2, &spins2};
    hipKernelNodeParams kp{}; kp.func = (void *)k_work; kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.kernelParams = args;
    t0 = now_us(); host = 0;
    for (int i = 0; i < N; ++i) {
        const double a = now_us();
        for (size_t k = 0; k < nn; ++k) hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp);
        hipGraphLaunch(ge, st);
        host += now_us() - a;
    }
    hipStreamSynchronize(st);
    printf("graph + new parameters per chain: host %.1f us per chain, elapsed %.1f us per chain (%zu nodes)\n", host / N, (now_us() - t0) / N, nn);
    return 0;
}

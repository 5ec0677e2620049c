// tools/ubench/lat.hip -- single-wavefront latency of the constructs the Viterbi sweep is made of (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o lat tools/ubench/lat.hip ; run: ./lat   (prints shader cycles per iteration)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_IT 2048

template <int CTRL> __device__ __forceinline__ int dppmin(int v) { return min(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true)); }

// every test: out[test] = cycles for N_IT iterations, one wave of 64 lanes (launch with `waves` wavefronts per block to load the SIMDs)
__global__ void k_lat(uint64_t *out, int *sink, int P, int test) {
    __shared__ int lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *L = lds + wave * 512;
    for (int i = lane; i < 512; i += 64) L[i] = i * 7 + P;
    __syncthreads();
    int x = lane, acc = P;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (test == 0) {          // empty loop (loop overhead: s_add, s_cmp, taken branch)
        for (int i = 0; i < N_IT; ++i) { asm volatile("" : "+v"(acc)); }
    } else if (test == 1) {   // dependent VALU chain of 8 adds
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc = acc * 3 + x; asm volatile("" : "+v"(acc)); }
        }
    } else if (test == 2) {   // LDS read -> dependent address -> read (pointer chase)
        for (int i = 0; i < N_IT; ++i) { acc = L[acc & 511]; }
    } else if (test == 3) {   // LDS write -> read back same location by another lane (the sweep's dp round trip)
        for (int i = 0; i < N_IT; ++i) { L[(lane + i) & 511] = acc; __builtin_amdgcn_wave_barrier(); acc += L[(lane + 1 + i) & 511]; }
    } else if (test == 4) {   // 3 dependent DPP mins
        for (int i = 0; i < N_IT; ++i) { acc = dppmin<0xB1>(acc); acc = dppmin<0x4E>(acc); acc = dppmin<0x141>(acc); acc += x; asm volatile("" : "+v"(acc)); }
    } else if (test == 5) {   // readlane -> scalar unpack -> scalar branch (descriptor dispatch)
        for (int i = 0; i < N_IT; ++i) {
            const uint32_t d = (uint32_t)__builtin_amdgcn_readlane(acc, i & 63);
            if (d >> 31) acc += 3; else acc += (d >> 17) & 127;
            asm volatile("" : "+v"(acc));
        }
    } else if (test == 6) {   // v_cmp -> exec-masked block with an LDS write (leaders)
        for (int i = 0; i < N_IT; ++i) {
            if ((lane & 7) == 0 && acc < P + (i << 20)) L[(lane + i) & 511] = acc;
            acc += x; asm volatile("" : "+v"(acc));
        }
    } else if (test == 7) {   // the sweep step in miniature: LDS reads (b64 + i16 + b32) -> add -> 3+3 DPP -> leaders write
        const uint2 *bk = (const uint2 *)L;
        for (int i = 0; i < N_IT; ++i) {
            const uint2 e = bk[(lane + i) & 127];
            const int pc = ((const short *)L)[(lane * 3 + i) & 1023];
            const int cs = L[(lane >> 3) + (i & 63)];
            int v = (int)e.x + pc;
            int vm = dppmin<0xB1>(v); vm = dppmin<0x4E>(vm); vm = dppmin<0x141>(vm);
            int nd = v == vm ? (int)(e.y >> 16) : 0x7FFFFFFF;
            nd = dppmin<0xB1>(nd); nd = dppmin<0x4E>(nd); nd = dppmin<0x141>(nd);
            if ((lane & 7) == 0) { L[(cs >> 16) & 511] = vm + (short)cs; ((short *)L)[600 + (lane >> 3) + (i & 63)] = (short)nd; }
            __builtin_amdgcn_wave_barrier();
            acc += vm;
        }
    } else if (test == 8) {   // global (L2-resident) dependent load chain: the trie walk / morph fetch
        for (int i = 0; i < N_IT; ++i) { acc = sink[(acc & 0xFFFF) * 16]; }
    } else if (test == 9) {   // 8 independent global gathers in flight, then use
        for (int i = 0; i < N_IT; ++i) {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += sink[((acc + k * 977 + lane * 131) & 0xFFFF) * 16];
            acc = s;
        }
    } else if (test == 10) {  // LDS atomic add with return
        for (int i = 0; i < N_IT; ++i) { acc += atomicAdd(&L[(lane + acc) & 511], 1); }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x12345678) sink[0] = acc;
}

int main() {
    uint64_t *d_out; int *d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 65536 * 16 * 4);
    std::vector<int> h(65536 * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i * 2654435761u) >> 7);
    hipMemcpy(d_sink, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char *names[] = {"empty loop", "8 dependent VALU (mul+add each)", "LDS pointer chase (read->read)", "LDS write->read round trip", "3 dependent DPP mins",
                           "readlane -> SALU unpack -> scalar branch", "v_cmp -> exec-masked LDS write", "sweep step in miniature", "global dependent load (4 MB table)",
                           "8 independent global gathers", "LDS atomic add with return"};
    for (int waves : {1, 4, 8, 16}) {
        printf("---- %d wavefront(s) per CU, 256 workgroups (cycles per iteration, wave 0 of block 0)\n", waves);
        for (int t = 0; t <= 10; ++t) {
            uint64_t best = ~0ull;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k_lat, dim3(256), dim3(64 * waves), 0, 0, d_out, d_sink, 5, t);
                uint64_t c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
                if (c < best) best = c;
            }
            printf("  %-44s %8.1f\n", names[t], (double)best / N_IT);
        }
    }
    return 0;
}

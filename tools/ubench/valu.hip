// tools/ubench/valu.hip -- the VALU issue rate of a wave64 integer op on gfx950: cycles per v_add_u32 per SIMD with 1 / 2 / 4 / 5 / 8 wavefronts
// per SIMD, independent accumulators (no dependency stalls).  Settles whether a wave64 VALU op occupies its SIMD for 2 or for 4 cycles -- and what
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU read for the same kernel (run once more under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
// SQ_BUSY_CYCLES --kernel-trace`; tools/valu_rate.sh does both).
// build: hipcc --offload-arch=gfx950 -O3 -o valu tools/ubench/valu.hip ; run: ./valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_IT 4096
#define OPS_PER_IT 64

// kind 0: v_add_u32 (one dword op), 1: v_mad_u32_u24 (three-operand), 2: v_min_i32 with a DPP row operand (what the sweep's minima are), 3: v_cndmask
__global__ void k_valu(uint64_t *out, uint32_t seed, int kind) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a0 = lane + seed, a1 = lane * 3 + seed, a2 = lane ^ seed, a3 = seed + 7, a4 = lane + 11, a5 = lane * 5, a6 = seed ^ 0x55, a7 = lane + 2 * seed;
    const uint32_t x = lane | 1;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int k = 0; k < OPS_PER_IT / 8; ++k) {
            if (kind == 0) {
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            } else if (kind == 1) {
                asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n"
                             "v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            } else if (kind == 2) {
                asm volatile("v_min_i32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_min_i32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_min_i32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_min_i32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            } else {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0) out[(size_t)blockIdx.x * 64 + wave] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345679u) out[0] = 0;
}

int main() {
    uint64_t *d_out;
    const int blocks = 256;
    hipMalloc(&d_out, blocks * 64 * 8);
    std::vector<uint64_t> h(blocks * 64);
    const char *names[] = {"v_add_u32", "v_mad_u32_u24", "v_min_i32 + DPP quad_perm", "v_cndmask_b32"};
    const double ops = (double)N_IT * OPS_PER_IT;
    printf("wave64 VALU ops, eight independent accumulators; a workgroup of 4 x W wavefronts = W per SIMD of one CU; %d workgroups; shader cycles (s_memtime)\n", blocks);
    printf("%-28s %5s %16s %22s\n", "instruction", "W", "cycles/op/wave", "cycles/op/SIMD (=/W)");
    for (int kind = 0; kind < 4; ++kind)
        for (int w : {1, 2, 4, 5, 8}) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256 * w), 0, 0, d_out, 5u + rep, kind);
                hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
                double mx = 0;
                for (int b = 0; b < blocks; ++b) for (int k = 0; k < 4 * w; ++k) mx = h[(size_t)b * 64 + k] > mx ? (double)h[(size_t)b * 64 + k] : mx;
                if (mx < best) best = mx;
            }
            printf("%-28s %5d %16.2f %22.2f\n", names[kind], w, best / ops, best / ops / w);
        }
    return 0;
}

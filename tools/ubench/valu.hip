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

// KIND 0: v_add_u32 (one dword op), 1: v_mad_u32_u24 (three operands), 2: v_min_i32 with a DPP quad_perm operand (what the sweep's minima are), 3: v_cndmask_b32.
// The loop body is ONE asm block of 64 instructions over eight independent accumulators (no branch, no dependency closer than eight instructions).
#define OP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define MAD(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %8\n"
#define MIN(k) "v_min_i32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define CND(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define CNDS(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
#define CMPCND(k) "v_cmp_lt_u32 vcc, %" #k ", %8\n v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define CNDADD(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n v_add_u32 %" #k ", %" #k ", %8\n v_add_u32 %" #k ", %" #k ", %8\n v_add_u32 %" #k ", %" #k ", %8\n"
#define ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %8\n"
#define LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define MINP(k) "v_min_i32 %" #k ", %" #k ", %8\n"
#define MOVDPP(k) "v_mov_b32_dpp %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define CMPS(k) "v_cmp_lt_u32_e64 s[10:11], %" #k ", %8\n v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
#define BODY(I) asm volatile(OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc", "s10", "s11")
template <int KIND>
__global__ void k_valu(uint64_t *out, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a0 = lane + seed, a1 = lane * 3 + seed, a2 = lane ^ seed, a3 = seed + 7, a4 = lane + 11, a5 = lane * 5, a6 = seed ^ 0x55, a7 = lane + 2 * seed;
    const uint32_t x = lane | 1;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_IT; ++i) {
        if constexpr (KIND == 0) BODY(ADD);
        else if constexpr (KIND == 1) BODY(MAD);
        else if constexpr (KIND == 2) BODY(MIN);
        else if constexpr (KIND == 3) BODY(CND);
        else if constexpr (KIND == 4) BODY(CNDS);
        else if constexpr (KIND == 5) BODY(CMPCND);
        else if constexpr (KIND == 6) BODY(CNDADD);
        else if constexpr (KIND == 7) BODY(ADD3);
        else if constexpr (KIND == 8) BODY(LSHLADD);
        else if constexpr (KIND == 9) BODY(AND);
        else if constexpr (KIND == 10) BODY(MINP);
        else if constexpr (KIND == 11) BODY(MOVDPP);
        else if constexpr (KIND == 12) BODY(MUL24);
        else BODY(CMPS);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0) out[(size_t)blockIdx.x * 64 + wave] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345679u) out[0] = 0;
}

template <int KIND>
static double run_kind(uint64_t *d_out, std::vector<uint64_t> &h, int blocks, int w) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256 * w), 0, 0, d_out, 5u + rep);
        if (hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        double mx = 0;
        for (int b = 0; b < blocks; ++b) for (int k = 0; k < 4 * w; ++k) mx = h[(size_t)b * 64 + k] > mx ? (double)h[(size_t)b * 64 + k] : mx;
        if (mx < best) best = mx;
    }
    return best;
}

int main() {
    uint64_t *d_out;
    const int blocks = 256;
    if (hipMalloc(&d_out, blocks * 64 * 8) != hipSuccess) return 1;
    std::vector<uint64_t> h(blocks * 64);
    // instructions per 64-slot body: 64, except the two-instruction (x2) and four-instruction (x4) patterns
    const char *names[] = {"v_add_u32 (VOP2)", "v_mad_u32_u24 (VOP3, 3 operands)", "v_min_i32 + DPP quad_perm", "v_cndmask_b32 ..., vcc (VOP2)", "v_cndmask_b32_e64 ..., s[10:11]",
                           "v_cmp_lt_u32 vcc + v_cndmask vcc", "v_cndmask vcc + 3 x v_add_u32", "v_add3_u32 (VOP3)", "v_lshl_add_u32 (VOP3)", "v_and_b32 (VOP2)",
                           "v_min_i32 (VOP2)", "v_mov_b32 + DPP quad_perm", "v_mul_u32_u24 (VOP2)", "v_cmp_e64 s[10:11] + v_cndmask_e64"};
    const int per_slot[] = {1, 1, 1, 1, 1, 2, 4, 1, 1, 1, 1, 1, 1, 2};
    printf("wave64 VALU ops, eight independent accumulators, one 64-slot asm body per loop iteration; a workgroup of 4 x W wavefronts = W per SIMD of one CU; %d workgroups; shader cycles (s_memtime)\n", blocks);
    printf("%-36s %3s %20s %24s\n", "instruction", "W", "cycles/instr/wave", "cycles/instr/SIMD (=/W)");
    for (int kind = 0; kind < 14; ++kind)
        for (int w : {1, 2, 4}) {
            double best = -1;
            switch (kind) {
                case 0: best = run_kind<0>(d_out, h, blocks, w); break; case 1: best = run_kind<1>(d_out, h, blocks, w); break;
                case 2: best = run_kind<2>(d_out, h, blocks, w); break; case 3: best = run_kind<3>(d_out, h, blocks, w); break;
                case 4: best = run_kind<4>(d_out, h, blocks, w); break; case 5: best = run_kind<5>(d_out, h, blocks, w); break;
                case 6: best = run_kind<6>(d_out, h, blocks, w); break; case 7: best = run_kind<7>(d_out, h, blocks, w); break;
                case 8: best = run_kind<8>(d_out, h, blocks, w); break; case 9: best = run_kind<9>(d_out, h, blocks, w); break;
                case 10: best = run_kind<10>(d_out, h, blocks, w); break; case 11: best = run_kind<11>(d_out, h, blocks, w); break;
                case 12: best = run_kind<12>(d_out, h, blocks, w); break; default: best = run_kind<13>(d_out, h, blocks, w); break;
            }
            const double ops = (double)N_IT * OPS_PER_IT * per_slot[kind];
            printf("%-36s %3d %20.2f %24.2f\n", names[kind], w, best / ops, best / ops / w);
        }
    return 0;
}

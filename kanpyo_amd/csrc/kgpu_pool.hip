// kanpyo_amd/csrc/kgpu_pool.hip -- the LDS-resident fused tokenize kernel (gfx950).
//
// One 64-lane wavefront owns one sentence from bytes to tokens; the whole lattice of the
// sentence lives in the CU's LDS (160 KB per CU on MI355X) and the Viterbi dependency chain
// touches nothing but LDS and registers:
//
//   * one double-array walk per start position; its matches (trie id, char length) are parked
//     in an LDS match buffer so the emit phase re-walks nothing (trie/da.rs:155-182 once per
//     position);
//   * every connection cost the sweep will need -- M[right(j)][left(t)] for each (target t,
//     predecessor j) pair, connection.rs:12-14 -- depends only on the morph ids, not on the DP
//     values, so all of them are gathered from HBM/L2 into an LDS pair table in ONE parallel pass
//     before the sweep ("the connection matrix tiled through LDS"); the sweep itself then runs
//     at LDS latency;
//   * per position the (target, predecessor) pairs are spread across the 64 lanes, each target
//     owning an aligned power-of-two lane group, and the strict-'<' first-minimum of
//     lattice.rs:125-139 is a DPP butterfly min-reduction on the 64-bit key
//     (total ^ signbit, predecessor node index).
//
// LDS page pool.  A workgroup is W independent wavefronts sharing one pool of 64 pages (a u64
// bitmap, first-fit runs of contiguous pages, LDS atomics).  Every wavefront reserves what its
// sentence is expected to need (host-adapted bytes-per-input-byte estimate), gives back what it
// does not use once the lattice is known, and releases the rest when the tokens are out.
// Occupancy therefore follows the sentences (short ones run 16 per CU, a long one may take a
// whole pool) instead of a launch-wide worst case, and no sentence that fits a pool waits for a
// later launch.
//
// Waiting rule (deadlock freedom): a wavefront waits for pages only while it holds none.  If
// its reservation turns out too small it releases it, waits for the exact size (now known)
// and redoes the sentence; a sentence that cannot fit an empty pool, or has more than MAXM
// dictionary prefixes at one position, goes to the next launch's work list.  The wavefronts of
// a workgroup never meet at a barrier after the pool is set up.
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "kgpu_device.h"

namespace kgpu {

using namespace dev;

namespace {

constexpr uint32_t MAXM = 8;         // trie matches buffered per start position
constexpr uint32_t NONE16 = 0xFFFFu;
#ifndef KGPU_POOL_SLEEP
#define KGPU_POOL_SLEEP 16
#endif
#ifndef KGPU_EST_SLACK
#define KGPU_EST_SLACK 768
#endif

// ---- DPP butterfly: min over aligned groups of 2^lg lanes, every lane gets it.
// The key is one u64 (total ^ signbit) << 32 | predecessor node index, so one
// v_cmp_lt_u64 + two v_cndmask per step, no branches.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_min_step(uint64_t k) {
    const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), CTRL, 0xF, 0xF, true);
    const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, CTRL, 0xF, 0xF, true);
    const uint64_t o = ((uint64_t)oh << 32) | ol;
    return o < k ? o : k;
}
__device__ __forceinline__ uint64_t shfl_min_step(uint64_t k, int d) {
    const uint32_t oh = (uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), d, 64);
    const uint32_t ol = (uint32_t)__shfl_xor((int)(uint32_t)k, d, 64);
    const uint64_t o = ((uint64_t)oh << 32) | ol;
    return o < k ? o : k;
}
__device__ __forceinline__ uint64_t group_min(uint64_t k, uint32_t lg) {  // lg wave-uniform
    if (lg >= 1) k = dpp_min_step<0xB1>(k);   // quad_perm [1,0,3,2]
    if (lg >= 2) k = dpp_min_step<0x4E>(k);   // quad_perm [2,3,0,1]
    if (lg >= 3) k = dpp_min_step<0x141>(k);  // row_half_mirror
    if (lg >= 4) k = dpp_min_step<0x140>(k);  // row_mirror
    if (lg >= 5) k = shfl_min_step(k, 16);
    if (lg >= 6) k = shfl_min_step(k, 32);
    return k;
}

__device__ __forceinline__ uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) & ~(a - 1); }

// ---- LDS page pool: 64 pages, bit i of *bm set = page i taken ---------------------------------
#ifdef KGPU_STEP_TIMING
constexpr uint32_t POOL_HDR = 32;  // + sum of the wavefronts' exit times, exit counter (idle time of finished wavefronts inside a live workgroup)
#else
constexpr uint32_t POOL_HDR = 16;
#endif
constexpr uint32_t POOL_PAGES = 64;
__device__ __forceinline__ uint64_t run_mask(uint32_t k, uint32_t pos) { return (k >= 64 ? ~0ull : ((1ull << k) - 1)) << pos; }

// One attempt to take k contiguous pages (first fit).  Wave-uniform result: page index or NONE.
__device__ __forceinline__ uint32_t pool_try_alloc(uint64_t *bm, uint32_t k, uint32_t lane) {
    for (int tries = 0; tries < 4; ++tries) {
        const uint64_t cur = bcast64(__hip_atomic_load(bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        uint64_t r = ~cur;  // bit i: pages i .. i+m-1 free
        for (uint32_t m = 1; m < k;) { const uint32_t t = min(m, k - m); r &= r >> t; m += t; }
        if (r == 0) return NONE;
        const uint32_t pos = (uint32_t)__ffsll((unsigned long long)r) - 1;
        bool won = false;
        if (lane == 0) won = atomicCAS((unsigned long long *)bm, (unsigned long long)cur, (unsigned long long)(cur | run_mask(k, pos))) == cur;
        if (__ballot(won) != 0) return pos;
    }
    return NONE;
}
__device__ __forceinline__ void pool_free(uint64_t *bm, uint32_t pos, uint32_t from, uint32_t to, uint32_t lane) {  // pages [pos+from, pos+to)
    wave_sync();
    if (lane == 0 && to > from) atomicAnd((unsigned long long *)bm, ~(unsigned long long)run_mask(to - from, pos + from));
}
// Wait (holding nothing) until k pages are free.  Bounded: gives up with NONE after ~0.5 s.
__device__ __forceinline__ uint32_t pool_wait_alloc(uint64_t *bm, uint32_t k, uint32_t lane) {
    for (uint32_t spin = 0; spin < (1u << 20); ++spin) {
        const uint32_t pg = pool_try_alloc(bm, k, lane);
        if (pg != NONE) return pg;
        __builtin_amdgcn_s_sleep(KGPU_POOL_SLEEP);
    }
    return NONE;
}


// The backtrace (lattice.rs:144-153), one lane: node `last` (EOS), then best predecessors (the low half of node[].y; a_y = LDS address of node[0].y) until a
// node has none -- BOS, or a node nothing reached; the indices go to the half-words at a_path, last first.  -> their count.  The walk costs the whole
// wavefront a slot of four cycles per instruction for one lane's work, K times a sentence, and the compiler's version of this loop is seventeen instructions a
// step (exec-mask bookkeeping for the divergent exit, the loaded half-word zero-extended again, values moved between registers): spelled out it is the
// address (one shift-add), the read, the test, the exit branch, the path's store -- two steps a round so that nothing moves, then the cursor, the count and
// the bound (node indices fall strictly along a path, so it ends within C + 1 steps; the walk is bounded all the same -- a round may run one step over: the
// two bytes behind path[] are the dead category array's -- and the count clamped).
__device__ __forceinline__ uint32_t backtrace_path(uint32_t a_y, uint32_t a_path, uint32_t last, uint32_t C) {
    uint32_t pos = last, pa = a_path, ad, pr, k;
    asm volatile(
        "s_mov_b32 %[k], 0\n"
        "Lbt_loop%=:\n\t"
        "v_lshl_add_u32 %[ad], %[pos], 3, %[ay]\n\t"
        "ds_read_u16 %[pr], %[ad]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32_e32 vcc, 0xffff, %[pr]\n\t"
        "s_cbranch_vccnz Lbt_done%=\n\t"
        "ds_write_b16 %[pa], %[pos]\n\t"
        "v_lshl_add_u32 %[ad], %[pr], 3, %[ay]\n\t"
        "ds_read_u16 %[pos], %[ad]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32_e32 vcc, 0xffff, %[pos]\n\t"
        "s_cbranch_vccnz Lbt_odd%=\n\t"
        "ds_write_b16 %[pa], %[pr] offset:2\n\t"
        "v_add_u32_e32 %[pa], 4, %[pa]\n\t"
        "s_add_u32 %[k], %[k], 2\n\t"
        "s_cmp_le_u32 %[k], %[c]\n\t"
        "s_cbranch_scc1 Lbt_loop%=\n\t"
        "s_branch Lbt_done%=\n"
        "Lbt_odd%=:\n\t"
        "s_add_u32 %[k], %[k], 1\n"
        "Lbt_done%=:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [k] "=&s"(k), [pos] "+v"(pos), [pa] "+v"(pa), [ad] "=&v"(ad), [pr] "=&v"(pr)
        : [ay] "s"(a_y), [c] "s"(C)
        : "vcc", "scc", "memory");
    return min(k, C + 1);
}

}  // namespace

// PROF: device-side work counters + per-phase shader clocks (KGPU_PROFILE_WORK).  A separate
// instantiation, because the 16 wave-uniform u64 accumulators + 9 ticks cost ~50 of the 102 SGPRs
// and push the plain kernel into SGPR spilling (v_readlane / v_writelane traffic on the VALU).
#ifndef KGPU_POOL_WPE
#define KGPU_POOL_WPE 4
#endif
// The arguments stay in the kernarg segment and every phase reads the fields it needs from there (scalar loads,
// KGPU_ARGS() at the phase boundaries): as ordinary by-value parameters the ~85 dwords are loaded at entry, live to
// the end, and -- 102 SGPRs per wavefront -- spilled to VGPR lanes and read back with v_readlane all over the kernel,
// on the VALU, once per sentence.
struct PoolArgs {
    DictView d; BatchArgs a; WorkIO io;
    uint32_t pool_bytes, max_pages, stop_after /* ablation timing only; 0 = run everything */;
    uint32_t page, page_magic;   // bytes per page of the pool (pool_page_bytes) and floor(2^32 / page) + 1: pages_for is one multiply-high (exact below 2^32 / page bytes)
};
// BYTE: the dictionary has no character-level copy of its trie (KGPU_BYTE_TRIE, or a key set it cannot represent): the walk goes byte by byte.
// A separate instantiation chosen at launch: the product kernel carries one walker, not two (SGPRs, spill code, instruction cache).
// WIDE: a parked match is two words {trie id, chars | records << 8} instead of one (a dictionary of 2^21 records or more, or leaves without their record
// count): likewise chosen at launch -- tested per match in the walk and in the emit phase it was four instructions each time.
template <bool PROF, bool BYTE, bool WIDE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(KGPU_POOL_WPE))) void k_tokenize_pool(PoolArgs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pool[];
    typedef const __attribute__((address_space(4))) PoolArgs *KArgs;
    const KArgs kargs = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    DictView d; BatchArgs a; WorkIO io;
    uint32_t stop_after;
    // (the empty asm makes the pointer opaque: what was read before it is dead, what is not used before the next one is never loaded)
#define KGPU_ARGS() do { KArgs kq_ = kargs; asm volatile("" : "+s"(kq_)); \
        d.da = kq_->d.da; d.da_len = kq_->d.da_len; d.leaf_dup = kq_->d.leaf_dup; d.first = kq_->d.first; d.morph = kq_->d.morph; d.n_morph = kq_->d.n_morph; d.unk_morph = kq_->d.unk_morph; d.n_unk_morph = kq_->d.n_unk_morph; d.conn = kq_->d.conn; d.conn_rows = kq_->d.conn_rows; d.bos_right = kq_->d.bos_right; d.eos_left = kq_->d.eos_left; d.cat = kq_->d.cat; d.cat_len = kq_->d.cat_len; d.cinfo = kq_->d.cinfo; d.da2 = kq_->d.da2; d.da2_len = kq_->d.da2_len; d.n_nb = kq_->d.n_nb; d.crec = kq_->d.crec; d.nb_cp = kq_->d.nb_cp; d.nb_code = kq_->d.nb_code; \
        a.utf8 = kq_->a.utf8; a.offsets = kq_->a.offsets; a.n = kq_->a.n; a.ctl = kq_->a.ctl; a.arena = kq_->a.arena; a.arena_bytes = kq_->a.arena_bytes; a.stage = kq_->a.stage; a.tok_count = kq_->a.tok_count; a.status = kq_->a.status; a.out = kq_->a.out; a.out_cap = kq_->a.out_cap; a.tok_offsets = kq_->a.tok_offsets; a.count_work = kq_->a.count_work; a.ovf[0] = kq_->a.ovf[0]; a.ovf[1] = kq_->a.ovf[1]; a.ovf[2] = kq_->a.ovf[2]; a.ovf[3] = kq_->a.ovf[3]; a.est_q8 = kq_->a.est_q8; a.dump_lattice = kq_->a.dump_lattice; a.fused_host = kq_->a.fused_host; a.fused_seq = kq_->a.fused_seq; a.stat_slots = kq_->a.stat_slots; \
        io.in_list = kq_->io.in_list; io.in_count = kq_->io.in_count; io.out_list = kq_->io.out_list; io.out_count = kq_->io.out_count; io.late_count = kq_->io.late_count; stop_after = kq_->stop_after; } while (0)
    KGPU_ARGS();
    const uint32_t max_pages = kargs->max_pages;
    const uint32_t lane = threadIdx.x & 63u, wave = bcast32(threadIdx.x >> 6) /* SGPR: everything per-sentence is wave-uniform */, W = blockDim.x >> 6;
    const int32_t base_root = d.da[1].base;
    // a sentence this kernel does not serve: onto the next launch's list; its token count reads 0 until a later kernel has served it (when the
    // host left the tail of the chain out -- no recent batch needed it -- the batch is run again with it: kgpu_api.cpp)
    auto defer_s = [&](uint64_t s_) { work_defer(io, lane, s_); if (lane == 0) a.tok_count[s_] = 0; };
    uint64_t *bm = (uint64_t *)pool;
    // W == 1: the workgroup is one wavefront with a pool of its own -- a fixed LDS slice.  Nothing to share, nothing to wait for:
    // every sentence gets the whole slice (no estimate, no redo), what does not fit goes to the next launch.
    const bool own_slice = W == 1;
    const uint32_t page = kargs->page, page_magic = kargs->page_magic;
    auto pages_for = [&](uint32_t bytes) { return __umulhi(bytes + page - 1, page_magic); };   // = (bytes + page - 1) / page (launch_tokenize_pool)
    // the workgroup's sentences -- list entries blockIdx.x + k * gridDim.x -- are handed to its wavefronts by a ticket in LDS (the dword behind the bitmap): a
    // wavefront that is through takes the next one, instead of every wavefront owning every W-th (with four or more sentences per wavefront -- batches of 16 384 and
    // more -- the static form left the early finishers of a workgroup waiting for its slowest wavefront's whole share; no global atomic: that lost in round 2)
    uint32_t *ticket = (uint32_t *)(pool + 8);
    if (threadIdx.x == 0) { *bm = 0; *ticket = 0; }
#ifdef KGPU_STEP_TIMING
    if (threadIdx.x == 0) { *(uint64_t *)(pool + 16) = 0; *(uint32_t *)(pool + 24) = 0; }
    const uint64_t tm_w0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();  // the only workgroup barrier: from here on the wavefronts are independent
    // profiling accumulators of this workgroup (flushed once at exit: per-sentence
    // atomics on a handful of hot words distort what they measure)
    uint64_t accW[7] = {0, 0, 0, 0, 0, 0, 0}, accP[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef KGPU_STEP_TIMING  // measurement build only (make timing): s_memtime ticks of the sweep, its slow-path steps, the whole sentence, the wait for pages
    uint64_t tmS = 0, tmSteps = 0, tmSlow = 0, tmSlowSteps = 0, tmSent = 0, tmPool = 0, tmN = 0, tmDesc = 0;
    uint64_t tmPh[7] = {0, 0, 0, 0, 0, 0, 0};  // load, decode, walk, scan, emit, gather + sweep, backtrace + tokens
#define KGPU_TM(...) __VA_ARGS__
#else
#define KGPU_TM(...)
#endif

    for (uint32_t iter = 0;; ++iter) {
        KGPU_ARGS();
        uint64_t s = 0;
        // sentence i of the work list -> workgroup i mod G, wavefront (i / G) mod W
#ifdef KGPU_POOL_STATIC   // measurement build: sentence i of the work list -> workgroup i mod G, wavefront (i / G) mod W, as before round 5
        if (!work_next_at(io, a, (uint64_t)blockIdx.x + (uint64_t)gridDim.x * (wave + (uint64_t)W * iter), s)) break;
#else
        (void)iter;
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(ticket, 1u);
        tk = bcast32(tk);
        // ticket tk of workgroup b -> list entry (tk / W) * (G W) + b W + tk % W: W CONSECUTIVE entries per workgroup and round -- neighbours in the text (their
        // offsets and bytes share cache lines) and, when the list is ordered by length (k_order_by_length), sentences that finish together: a workgroup keeps its
        // LDS until its last wavefront is through
        // (a workgroup's first W tickets -- all of them in a launch of one sentence per wavefront -- need no division by the run-time W)
        const uint64_t widx = tk < W ? (uint64_t)blockIdx.x * W + tk : (uint64_t)(tk / W) * ((uint64_t)gridDim.x * W) + (uint64_t)blockIdx.x * W + tk % W;
        if (!work_next_at(io, a, widx, s)) break;
#endif
        const uint64_t b0 = a.offsets[s];
        const uint64_t Bl = a.offsets[s + 1] - b0;
        const uint32_t pool_cap = page * POOL_PAGES;
        if (Bl + 64 > pool_cap || Bl > 0xFFF0) { defer_s(s); continue; }
        const uint32_t B = (uint32_t)Bl;
        const uint8_t *gtext = a.utf8 + b0;
        // chars of the sentence (sizes the per-position arrays).  A sentence of up to ~250 bytes is ONE load: lane l takes the aligned dword l of the text (an
        // aligned dword that holds a byte of the sentence never leaves its page), a byte-align with the next lane's word makes it bytes 4 l .. 4 l + 3 of the
        // sentence, bytes past the end become continuation bytes (0x80: they start no character, and the four behind the end are the padding the decoder
        // reads); the characters are counted from the words and the words go to LDS as they are below.  Longer ones: a byte per lane and round, re-read from L1.
        const uint32_t tsh = (uint32_t)(uintptr_t)gtext & 3u;
        const bool one_load = B + tsh + 4 <= 256;
        auto load_words = [&]() {
            uint32_t w = 0x80808080u;
            const uint32_t lo = 4 * lane;
            if (B != 0 && lo < tsh + B) w = *(const uint32_t *)(gtext - tsh + lo);
            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)w, (int)w, 0x130 /* wave_shl:1: lane l reads lane l + 1 */, 0xF, 0xF, false);
            const uint32_t al = __builtin_amdgcn_alignbyte(nx, w, tsh);
            const uint32_t q1 = (uint32_t)min(max((int32_t)B - (int32_t)lo, 0), 4);
            const uint32_t m = q1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * q1)) - 1u);   // the sentence's bytes in this word
            return (al & m) | (0x80808080u & ~m);
        };
        uint32_t C = 0, tw = 0;
        if (one_load) {
            tw = load_words();
            const uint32_t cont = tw & ~(tw << 1) & 0x80808080u;   // bit 7 of every 10xxxxxx byte
            C = 256u - (uint32_t)(__popcll(__ballot((cont & 0x80u) != 0)) + __popcll(__ballot((cont & 0x8000u) != 0)) + __popcll(__ballot((cont & 0x800000u) != 0)) + __popcll(__ballot((cont & 0x80000000u) != 0)));
        } else
            for (uint32_t k0 = 0; k0 < B; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t b = k < B ? gtext[k] : 0x80u;
                C += __popcll(__ballot(k < B && (b & 0xC0) != 0x80));
            }
        // reservation: what the per-position arrays + match buffer need for sure, or the host-adapted
        // estimate of the whole lattice, whichever is larger.  A sentence expected not to fit an empty
        // pool is routed on without paying for a trie walk that would be thrown away.
        // a parked match: {trie id, chars | records << 8}; one word id (21 bits) | chars (8) | records (3; 0 = look the count up)
        // when the ids fit (DictView::leaf_dup: fewer than 2^21 morphs)
        constexpr uint32_t MS = WIDE ? 8u : 4u;   // (launch_tokenize_pool: one word when d.leaf_dup and fewer than 2^21 records)
        const uint32_t need1 = align_up(B + 4, 4) + 22 * (C + 2) + 2 * align_up(C + 2, 4) + align_up(C * MAXM * MS, 16) + 32;
        const uint32_t est = max(need1, (uint32_t)(((uint64_t)B * a.est_q8) >> 8) + KGPU_EST_SLACK);
        uint32_t npg = own_slice ? POOL_PAGES : pages_for(est);
        // routing: a sentence expected to need more than max_pages would hold a large part of the pool for a long
        // time (LDS x time grows with the square of the length); it is better served by the long-sentence kernel
        if (own_slice ? need1 > pool_cap : npg > max_pages) { defer_s(s); continue; }
        KGPU_TM(const uint64_t tm_p0 = __builtin_amdgcn_s_memtime();)
        uint32_t pg = pool_wait_alloc(bm, npg, lane);
        KGPU_TM(const uint64_t tm_p1 = __builtin_amdgcn_s_memtime(); tmPool += tm_p1 - tm_p0;)
        if (pg == NONE) { defer_s(s); continue; }
        for (uint32_t attempt = 0;; ++attempt) {  // at most one redo, with the exact size
        uint8_t *smem = pool + POOL_HDR + pg * page;
        const uint32_t lds_bytes = npg * page;

        uint64_t tick[9];
#ifdef KGPU_STEP_TIMING
        constexpr bool prof = true;
#else
        constexpr bool prof = PROF;
#endif
#define KGPU_TICK(k) do { if (prof) tick[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define KGPU_STOP(k) if (stop_after == (k)) { if (lane == 0) { a.status[s] = KGPU_SENT_TRUNCATED; a.tok_count[s] = 0; } break; }
        KGPU_ARGS();
        KGPU_TICK(0);
        // ---- phase 0a: stage the sentence in LDS, count chars -----------------
        uint8_t *text = smem;
        if (one_load && attempt == 0) {   // (a redo stages the bytes the long way)
            if (4 * lane < B + 4) *(uint32_t *)(smem + 4 * lane) = tw;
        } else
            for (uint32_t k0 = 0; k0 < B + 4; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t b = k < B ? gtext[k] : 0x80u;
                if (k < B + 4) text[k] = (uint8_t)b;
            }
        // ---- LDS carve: per-char arrays from the bottom, match buffer from the top
        uint32_t off = align_up(B + 4, 4);
        uint32_t *nb = (uint32_t *)(smem + off);    off += 4 * (C + 2);  // node count -> first node index
        uint32_t *boff = (uint32_t *)(smem + off);  off += 4 * (C + 2);  // bucket count -> offset (edges[e])
        uint32_t *bfill = (uint32_t *)(smem + off); off += 4 * (C + 2);  // bucket fill cursor
        uint32_t *ebase = (uint32_t *)(smem + off); off += 4 * (C + 2);  // first pair index per position
        uint16_t *cbyte = (uint16_t *)(smem + off); off += 2 * (C + 2);  // char -> byte offset
        uint16_t *uspan = (uint16_t *)(smem + off); off += 2 * (C + 2);  // unknown span (0 = none)
        uint16_t *cp16 = (uint16_t *)(smem + off);  off += 2 * (C + 2);  // BMP code point (0xFFFF: not BMP), then the character's code: the walk's
        uint16_t *path = cp16;                                            // ... and, once the walk is through, the backtrace's
        uint8_t *ccat = smem + off;                 off += align_up(C + 2, 4);
        uint8_t *mcnt = smem + off;                 off += align_up(C + 2, 4);
        const uint32_t mbytes = align_up(C * MAXM * MS, 16);
        // off + mbytes <= need1 <= lds_bytes by construction of the reservation
        const uint32_t moff = (lds_bytes - mbytes) & ~15u;
        uint32_t *mbuf = (uint32_t *)(smem + moff);          // [C][MAXM] parked matches
        wave_sync();
        KGPU_TICK(1);
        KGPU_STOP(1)
        KGPU_ARGS();

        // ---- phase 0b: decode + validate + category --------------------------------
        uint32_t cb = 0, bad = 0, lensum = 0;
        for (uint32_t k0 = 0; k0 < B; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t b = k < B ? text[k] : 0x80u;
            const bool start = k < B && (b & 0xC0) != 0x80;
            const uint64_t m = __ballot(start);
            const uint32_t ci = cb + __popcll(m & ((1ull << lane) - 1));
            if (start) {
                uint32_t cp;
                const uint32_t l = utf8_decode_lead(b, k, B, [&](uint32_t kk) -> uint32_t { return text[kk]; }, cp, bad);
                lensum += l;
                cbyte[ci] = (uint16_t)k;
                cp16[ci] = (uint16_t)(cp < 0xFFFFu ? cp : 0xFFFFu);
                // the category of a BMP character is loaded with the walk's first loads (phase 1), not here: this phase then has no
                // exposed trip to the category table.  Only what cp16 cannot name (non-BMP -> table[0], and U+FFFF) is looked up now.
                if (cp >= 0xFFFFu) ccat[ci] = bad ? 0 : (cp < d.cat_len ? d.cat[cp] : d.cat[0]);  // char_category_def.rs:33-38
            }
            cb += __popcll(m);
        }
        lensum = bcast32(wave_sum(lensum));  // keep every early exit wave-uniform (SGPR) for the compiler
        if (__ballot(bad != 0) != 0 || lensum != B) {
            if (lane == 0) { a.status[s] = KGPU_SENT_INVALID_UTF8; a.tok_count[s] = 0; }
            break;
        }
        if (lane == 0) cbyte[C] = (uint16_t)B;
        for (uint32_t e = lane; e < C + 2; e += 64) { boff[e] = 0; bfill[e] = 0; }
        wave_sync();

        KGPU_TICK(2);
        KGPU_STOP(2)
        KGPU_ARGS();
        // ---- phase 1: one trie walk per start position; count + park matches ------
        uint32_t wT = 0, ovf = 0;
        {
            const int nchunks = (int)((C + 63) / 64);
            uint32_t carry_end = C;
            for (int ch = nchunks - 1; ch >= 0; --ch) {
                const uint32_t i = (uint32_t)ch * 64 + lane;
                const bool active = i < C;
                const uint32_t cpi = active ? cp16[i] : 0xFFFFu;
                uint32_t cat = 0x1FFu, ncat = 0x2FFu;
                uint32_t cnt = 0, m = 0;
                auto on_match = [&](uint32_t id, uint32_t nch, uint32_t dup) {
                    const uint32_t nrec = 1u + (dup != NONE ? dup : (uint32_t)d.morph[id - 1].dup);  // index.rs:46-51
                    if (m < MAXM && nch < 256) {
                        if (MS == 4) mbuf[i * MAXM + m] = id | (nch << 21) | ((nrec < 8 ? nrec : 0u) << 29);
                        else *(uint2 *)(mbuf + 2 * (i * MAXM + m)) = make_uint2(id, nch | (nrec << 8));
                    } else ovf = 1;
                    ++m;
                    cnt += nrec;
                    atomicAdd(&boff[i + nch], nrec);
                };
                if constexpr (!BYTE) {
                    // Character-level array (kgpu_chartrie.cpp): ONE 16-byte record per character gives its category, its code and the
                    // root's child for it; the codes replace the code points in LDS (the chunks are walked last to first, so the
                    // characters a walk runs into already have theirs), then every further character costs one dependent load.
                    const CharRec r = d.crec[cpi != 0xFFFFu ? cpi : 0u];
                    uint32_t code = 0xFFFFu;
                    int32_t p0 = 0, bp0 = 0, lf0 = 0;
                    if (active) {
                        if (cpi != 0xFFFFu) { cat = r.cat; code = r.code; p0 = r.slot; bp0 = r.base; lf0 = r.leaf; }
                        else {  // not in the table (>= U+FFFF): category from the decode phase, code from the dictionary's short list
                            cat = ccat[i];
                            code = d.n_nb ? ct_code_nonbmp(d, utf8_cp_at(text + cbyte[i])) : 0xFFFFu;
                            if (code != 0xFFFFu) {
                                const uint32_t q = (uint32_t)d.da2[1].base + code;
                                if (q < d.da2_len) { const CtNode nd = d.da2[q]; if (nd.check == 1) { p0 = (int32_t)q; bp0 = nd.base; lf0 = nd.leaf; } }
                            }
                        }
                        cp16[i] = (uint16_t)code;
                        ccat[i] = (uint8_t)cat;  // (the next position's run test and the emit phase's fallback read it)
                    }
                    wave_sync();
                    if (i + 1 < C) ncat = ccat[i + 1];
                    if (active) {
                        ct_walk(d, p0, bp0, lf0, [&](uint32_t dep) -> uint32_t { return i + dep < C ? (uint32_t)cp16[i + dep] : 0xFFFFu; }, on_match);
                        if constexpr (PROF) if (a.count_work == 1u) wT += da_walk_first(d, text, cpi, cbyte[i], cbyte[i + 1], B, base_root, [](uint32_t, uint32_t, uint32_t) {});  // the reference's byte steps (work counters)
                        mcnt[i] = (uint8_t)(m < MAXM ? m : MAXM);
                    }
                } else {
                // categories (char_category_def.rs:33-38) of this character and of the next: requested here, consumed after the walk
                const uint32_t cpn = (i + 1 < C) ? cp16[i + 1] : 0u;
                if (active) cat = cpi != 0xFFFFu ? (cpi < d.cat_len ? d.cat[cpi] : d.cat[0]) : ccat[i];
                if (i + 1 < C) ncat = cpn != 0xFFFFu ? (cpn < d.cat_len ? d.cat[cpn] : d.cat[0]) : ccat[i + 1];
                if (active) {
                    wT += da_walk_first(d, text, cpi, cbyte[i], cbyte[i + 1], B, base_root, on_match);
                    mcnt[i] = (uint8_t)(m < MAXM ? m : MAXM);
                    ccat[i] = (uint8_t)cat;  // (the emit phase's fallback reads it)
                }
                }
                const uint64_t bm = __ballot(active && ncat != cat);
                const uint64_t rest = bm >> lane;
                const uint32_t run_end = rest ? i + (uint32_t)__ffsll((unsigned long long)rest) : carry_end;
                carry_end = bcast32(run_end);
                if (active) {
                    const CatInfo ci = d.cinfo[cat];
                    uint32_t span = 0;
                    if ((cnt == 0 || (ci.flags & CAT_INVOKE)) && (ci.flags & CAT_HAS_UNK) && ci.unk_count) {  // lattice.rs:54,87-92
                        span = 1;
                        if (ci.flags & CAT_GROUP) {  // lattice.rs:66-84
                            const uint32_t r = run_end - i;
                            span = r < MAX_UNKNOWN_LEN ? r : MAX_UNKNOWN_LEN;
                        }
                        cnt += ci.unk_count;
                        atomicAdd(&boff[i + span], ci.unk_count);
                        if (m < MAXM) {  // the unknown records ride behind the matches: the emit phase reads no category table
                            if (MS == 4) mbuf[i * MAXM + m] = (uint32_t)ci.unk_first | ((ci.unk_count < 8 ? ci.unk_count : 0u) << 29);
                            else *(uint2 *)(mbuf + 2 * (i * MAXM + m)) = make_uint2((uint32_t)ci.unk_first, ci.unk_count);
                        }
                    }
                    uspan[i] = (uint16_t)span;
                    nb[i] = cnt;
                }
            }
        }
        if (__ballot(ovf != 0) != 0) { defer_s(s); break; }
        if (lane == 0) {
            nb[C] = 1;       // EOS starts at C (lattice.rs:165-175)
            nb[C + 1] = 0;
            atomicAdd(&boff[0], 1u);  // BOS ends at 0 (lattice.rs:156-164)
        }
        wave_sync();

        KGPU_TICK(3);
        KGPU_STOP(3)
        KGPU_ARGS();
        // ---- phase 2: prefix sums: node numbering, bucket offsets, tile offsets ------
        // A TILE is the unit of stage B: up to 8 targets x 8 predecessors of one start position, pair (ti, j) on lane 8 ti + j; a position with T
        // targets and P predecessors is ceil(T / 8) x max(1, ceil(P / 8)) tiles (P = 0: one tile of absent candidates -- every target stays at INF).
        uint32_t ncarry = 1, bcarry = 0, tcarry = 0;
        uint64_t Esum = 0;
        for (uint32_t i0 = 0; i0 < C + 2; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint32_t v = i < C + 2 ? nb[i] : 0;    // targets starting at i
            const uint32_t w = i < C + 2 ? boff[i] : 0;  // predecessors ending at i
            const uint32_t x = v ? ((v + 7u) >> 3) * max(1u, (w + 7u) >> 3) : 0u;  // tiles of position i
            const uint32_t vs = wave_incl_scan(v, lane), ws = wave_incl_scan(w, lane), xs = wave_incl_scan(x, lane);
            if (i < C + 2) { nb[i] = ncarry + vs - v; boff[i] = bcarry + ws - w; ebase[i] = tcarry + xs - x; }
            ncarry += __shfl(vs, 63, 64);
            bcarry += __shfl(ws, 63, 64);
            tcarry += __shfl(xs, 63, 64);
            if constexpr (PROF) Esum += (uint64_t)v * w;  // relaxations at i (lattice.rs:122-125): the work counter E
        }
        // scalarise: the carve and the fit test below must be wave-uniform branches
        const uint32_t N = bcast32(ncarry), Nb = bcast32(bcarry), NT = bcast32(tcarry);

        // ---- LDS carve, part 2: node arrays, buckets, tile list ---------------------
        // What emit 3a writes (per node: morph id, end position) lies BELOW the match buffer it reads; the buckets (filled by 3b) and the tile list (3c) may
        // lie over it: a sentence's peak is max(3a's, the sweep's), not their sum.  A node's start position is not kept: the tokens look it up in nb[].
        off = align_up(off, 8);
        uint2 *node = (uint2 *)(smem + off);        off += 8 * (N + 1);  // {word cost (i16) | bucket slot of the node << 16, byte offset of the node's matrix row (left * rows * 2)}; [N]: the padding tiles' target
                                                                    // the sweep stores the best predecessor into the low half of .y once the node's costs are gathered
        int32_t *nSid = (int32_t *)(smem + off);    off += 4 * N;   // +id known, -id unknown, 0 dummy
        off = align_up(off, 8);
        const uint32_t off_emit_end = off;                          // everything above is written by emit 3a
        uint2 *bk = (uint2 *)(smem + off);          off += 8 * (Nb + 2);  // bucket (= edges[e]): {dp, 2 * right | node << 16}; [Nb]: sink for EOS; [Nb + 1]: the absent candidate
        uint2 *tiles = (uint2 *)(smem + off);                       // the tile list
        if (N > 0xFFFF || Nb + 1 > SLOT_MAX) { defer_s(s); break; }   // (neither fits a pool: 8 bytes per bucket entry)
        // exact requirement: what 3a writes stays below the match buffer; afterwards the buckets and the tile list overlay it
        const uint32_t need_emit = off_emit_end + mbytes + 16, need_full = off + 8 * NT;
        if (need_emit > lds_bytes || need_full > lds_bytes) {
            // reservation too small: release, wait (holding nothing) for the exact size, redo
            pool_free(bm, pg, 0, npg, lane);
            if (lane == 0) atomicAdd(io.late_count, 1u);
            pg = NONE;
            if (pages_for(max(need_emit, need_full)) > max_pages || attempt != 0) { defer_s(s); break; }
            npg = pages_for(max(need_emit, need_full));
            pg = pool_wait_alloc(bm, npg, lane);
            if (pg == NONE) { defer_s(s); break; }
            continue;
        }
        wave_sync();

        KGPU_TICK(4);
        KGPU_STOP(4)
        KGPU_ARGS();
        // ---- phase 3: emit nodes from the parked matches --------------------------------
        // 3a, lane = start position, LDS only: the node list in insertion order (lattice.rs:177-201) -- per node its
        // morph id, start and (parked in node[].y) end position
        for (uint32_t i = lane; i < C; i += 64) {
            uint32_t t = nb[i];
            const uint32_t nm = mcnt[i], span = uspan[i];
            uint32_t ufirst = 0, ucnt = 0;
            if (span) {
                if (nm < MAXM) {
                    if (MS == 4) { const uint32_t w = mbuf[i * MAXM + nm]; ufirst = w & 0x1FFFFFu; ucnt = w >> 29; }
                    else { const uint2 w = *(const uint2 *)(mbuf + 2 * (i * MAXM + nm)); ufirst = w.x; ucnt = w.y; }
                }
                if (ucnt == 0) { const CatInfo ci = d.cinfo[ccat[i]]; ufirst = (uint32_t)ci.unk_first; ucnt = ci.unk_count; }  // no room, or eight or more
            }
            for (uint32_t m = 0; m < nm; ++m) {
                uint32_t id, end, nrec;
                if (MS == 4) {
                    const uint32_t w = mbuf[i * MAXM + m];
                    id = w & 0x1FFFFFu; end = i + ((w >> 21) & 255u); nrec = w >> 29;
                    if (nrec == 0) nrec = 1u + d.morph[id - 1].dup;  // eight or more records on one surface
                } else {
                    const uint2 w = *(const uint2 *)(mbuf + 2 * (i * MAXM + m));
                    id = w.x; end = i + (w.y & 255u); nrec = w.y >> 8;
                }
                for (uint32_t r = 0; r < nrec; ++r, ++t) { nSid[t] = (int32_t)(id + r); node[t].y = end; }
            }
            if (span)   // lattice.rs:87-97,190-201
                for (uint32_t r = 0; r < ucnt; ++r, ++t) { nSid[t] = -(int32_t)(ufirst + r); node[t].y = i + span; }
        }
        wave_sync();
        // 3b, lane = node: its morph record (one gather per 64 nodes instead of one dependent load per record of the
        // busiest position), its slot in the bucket of the position it ends at.  The order inside a bucket is free:
        // the sweep breaks ties on the node index it carries.
        const uint32_t rows2 = d.conn_rows * 2u;   // bytes per matrix row (connection.rs:12-14: element (right, left) at left * rows + right)
        for (uint32_t t0 = 1; t0 < N - 1; t0 += 256) {  // four nodes per lane: the four record gathers are in flight together
            uint32_t tt[4], ee[4];
            Morph8 mm[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tt[k] = t0 + 64 * k + lane;
                const uint32_t tc = min(tt[k], N - 2);   // (lanes past the last word read its entries: no exec mask around the reads; N >= 3 here)
                const int32_t sid = nSid[tc];
                ee[k] = node[tc].y;
                mm[k] = d.morph[sid > 0 ? (uint32_t)sid - 1u : d.n_morph - 1u - (uint32_t)sid];   // the unknown words' records follow the known ones' (DictView)
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (tt[k] < N - 1) {
                    const uint32_t slot = boff[ee[k]] + atomicAdd(&bfill[ee[k]], 1u);
                    node[tt[k]] = make_uint2((uint32_t)(uint16_t)mm[k].cost | (slot << SLOT_SHIFT), (uint32_t)(uint16_t)mm[k].left * rows2);
                    bk[slot].y = ((uint32_t)(uint16_t)mm[k].right << 1) | (tt[k] << 16);   // ids are non-negative i16 (checked at create): 2 * right fits the half-word
                }
            }
        }
        if (lane == 0) {
            node[N - 1] = make_uint2(Nb << SLOT_SHIFT, d.eos_left * rows2);  // EOS: Morph(0,0,0), ranked id; its dp goes to the sink slot
            nSid[N - 1] = 0;
            bk[0] = make_uint2(0u, d.bos_right << 1);  // BOS: dp None -> 0 (lattice.rs:127), right_id 0 (ranked), node 0
            bk[Nb + 1] = make_uint2(0x7FFEFFFFu, 0u);  // what a position without predecessors relaxes from: a total no real one reaches (real <= INF + 32767) that
                                                       // cannot overflow when a connection cost and a word cost are added, and stays >= INF when they are negative
            node[0] = make_uint2(0u, NONE16);          // ... and no predecessor (the backtrace stops here)
            node[N] = make_uint2(0u, 0u);              // what a padding tile gathers for: row 0 of the matrix, never swept, so never overwritten
        }
        wave_sync();
        {   // the match buffer is dead now: give back the pages beyond the tile list
            const uint32_t keep = pages_for(need_full);
            if (keep < npg) { pool_free(bm, pg, keep, npg, lane); npg = keep; }
        }
        const uint32_t lds0 = (uint32_t)(uintptr_t)(KGPU_LDS(uint8_t) *)pool;  // absolute LDS addresses, wave-uniform: SGPRs
        const uint32_t a_node = bcast32(lds0 + (uint32_t)((uint8_t *)node - pool)), a_bk = bcast32(lds0 + (uint32_t)((uint8_t *)bk - pool));
        // 3c, lane = start position: its tiles (the descriptors of kgpu_device.h: tile_desc0); (a, b) = (target group, predecessor chunk), b fastest.  A position
        // without predecessors (P = 0) is one chunk of one absent candidate (bk[Nb + 1]): its targets stay at INF with no predecessor.
        for (uint32_t q0 = 0; q0 <= C; q0 += 64) {
            const uint32_t q = q0 + lane;
            if (q <= C) {
                const uint32_t t0 = nb[q], T = nb[q + 1] - t0, p0r = boff[q], Pr = boff[q + 1] - p0r;
                const uint32_t p0 = Pr ? p0r : Nb + 1, P = Pr ? Pr : 1u;
                uint32_t k = ebase[q];
                const uint32_t kb = (P + 7u) >> 3;
                for (uint32_t ta = 0; ta < T; ta += 8)
                    for (uint32_t b = 0; b < kb; ++b, ++k)
                        tiles[k] = make_uint2(tile_desc0(a_node + 8 * (t0 + ta), min(8u, T - ta), min(8u, P - 8 * b), b == 0, b == kb - 1), a_bk + 8 * (p0 + 8 * b));
            }
        }
        const uint32_t null0 = (a_node + 8 * N) | TILE_FIRST;   // what the last group gathers for the tiles it does not have: a target group that is never reduced (M[BOS's right][0]: always in the matrix)
        wave_sync();
        KGPU_TICK(5);
        KGPU_STOP(5)
        KGPU_ARGS();
        uint64_t cyc_gather = 0;
        // ---- phase 4: stage B over the tile list (kgpu_device.h: tiles_run -- the gather into registers, the sweep; lattice.rs:116-142, connection.rs:12-14) ----
        {
            const uint64_t tg0 = prof ? __builtin_amdgcn_s_memtime() : 0;
            tiles_run(tiles, 0u, NT, make_uint2(null0, a_bk), lane, a_bk, (const uint8_t *)d.conn, stop_after != 6);   // (6: ablation timing -- the gathers alone)
            if (prof && stop_after == 6) cyc_gather += __builtin_amdgcn_s_memtime() - tg0;
        }
        wave_sync();

        KGPU_TICK(6);
        KGPU_STOP(6)
        KGPU_STOP(7)
        KGPU_ARGS();
        // ---- phase 5: backtrace (lattice.rs:144-153) + Node -> Token (tokenizer.rs:22-43)
        uint32_t K = 0;
        if (lane == 0) K = backtrace_path(a_node + 4, lds0 + (uint32_t)((uint8_t *)path - pool), N - 1, C);
        K = bcast32(K);
        // staging slot of the sentence: K <= C + 1 <= B + 1 tokens always fit at b0 + s
        // (no cursor atomics: a single hot word serialises ~90 sentences/us chip-wide)
        const uint64_t ts = b0 - a.offsets[0] + s;
        wave_sync();
        {
            // the start position of node t: the last i with nb[i] <= t (nb[] = first node index per position, ascending; an empty position shares its
            // successor's; nb[C + 1] = N > t stops a probe past the end) -- a binary search per token instead of a half-word per node written by emit and kept
            // in LDS through the sweep.  A word is never last on the path (EOS is): it ends where its successor starts -- the next lane's search.
            // (any power of two >= the highest one in C may start the search -- a probe past the end is clamped onto nb[C + 1] -- so up to 255 characters it is eight
            // steps with nothing between them; beyond, the loop)
            const uint32_t hb = 1u << (31 - __clz((int)max(C, 1u)));
            auto start_of = [&](uint32_t t) {
                uint32_t lo = 0;
                if (C < 256) {
#pragma unroll
                    for (uint32_t st = 128; st; st >>= 1) { const uint32_t m = min(lo + st, C + 1); if (nb[m] <= t) lo = m; }
                } else
                    for (uint32_t st = hb; st; st >>= 1) { const uint32_t m = min(lo + st, C + 1); if (nb[m] <= t) lo = m; }
                return lo;
            };
            for (uint32_t k0 = 0; k0 < K; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t t = path[K - 1 - min(k, K - 1)];
                const uint32_t st = start_of(t);
                uint32_t en = (uint32_t)__builtin_amdgcn_update_dpp((int)st, (int)st, 0x130 /* wave_shl:1: lane l reads lane l + 1 */, 0xF, 0xF, false);
                if (lane == 63 && k + 1 < K) en = start_of(path[K - 2 - k]);   // (more than 64 tokens: the successor belongs to the next round)
                if (k < K) {
                    const int32_t sid = nSid[t];
                    kgpu_token tk;
                    if (sid == 0) {  // Dummy -> "EOS" (tokenizer.rs:27-28,34)
                        tk.id = 0; tk.cls = KGPU_CLASS_DUMMY; tk.position = B; tk.start = C; tk.end = C + 3; tk.byte_len = 0;
                    } else {
                        const uint32_t bs = cbyte[st];
                        tk.id = sid > 0 ? sid : -sid;
                        tk.cls = sid > 0 ? KGPU_CLASS_KNOWN : KGPU_CLASS_UNKNOWN;
                        tk.position = bs; tk.start = st; tk.end = en; tk.byte_len = cbyte[en] - bs;
                    }
                    a.stage[ts + k] = tk;
                }
            }
        }
        if (lane == 0) { a.status[s] = KGPU_SENT_OK; a.tok_count[s] = K; }
        KGPU_TM(const uint64_t tm_e = __builtin_amdgcn_s_memtime(); tmSent += tm_e - tm_p1; tmN += 1;
                for (int k = 0; k < 6; ++k) tmPh[k] += tick[k + 1] - tick[k];
                tmPh[6] += tm_e - tick[6];)
        if constexpr (PROF) {
            wT = wave_sum(wT);
            const uint64_t t7 = __builtin_amdgcn_s_memtime();
            accW[0] += 1; accW[1] += B; accW[2] += C; accW[3] += wT; accW[4] += N - 1; accW[5] += bcast64(wave_sum64(Esum)); accW[6] += K;
            accP[0] += tick[1] - tick[0]; accP[1] += tick[2] - tick[1]; accP[2] += tick[3] - tick[2];
            accP[3] += tick[4] - tick[3]; accP[4] += tick[5] - tick[4]; accP[5] += cyc_gather;
            accP[6] += tick[6] - tick[5] - cyc_gather; accP[7] += t7 - tick[6]; accP[8] += 1;
        }
        break;
        }  // attempt
        if (pg != NONE) pool_free(bm, pg, 0, npg, lane);
    }
    KGPU_ARGS();
    {   // flush the profiling accumulators: the wavefront's own slot (plain adds, summed on the host)
#ifdef KGPU_STEP_TIMING
        constexpr bool flush = true;
        const uint64_t tm[8] = {tmS, tmSteps, tmSlow, tmSlowSteps, tmSent, tmPool, tmN, tmDesc};
        uint64_t tm_idle = 0, tm_life = 0;
        {
            const uint64_t te = __builtin_amdgcn_s_memtime();
            tm_life = te - tm_w0;
            uint32_t old = 0;
            if (lane == 0) { atomicAdd((unsigned long long *)(pool + 16), (unsigned long long)te); __threadfence_block(); old = atomicAdd((uint32_t *)(pool + 24), 1u); }
            old = bcast32(old);
            if (old == W - 1) tm_idle = (uint64_t)W * te - *(volatile uint64_t *)(pool + 16);  // the last one out: what the others waited
        }
#else
        constexpr bool flush = PROF;
#endif
        if (flush) {
            uint64_t v = 0;
#ifdef KGPU_STEP_TIMING
            if (!PROF) {
                for (int k = 0; k < 7; ++k) if (lane == (uint32_t)k) v = tmPh[k];
                for (int k = 0; k < 8; ++k) if (lane == 16u + k) v = tm[k];
                if (lane == 24u) v = tm_idle;
                if (lane == 25u) v = tm_life;
            }
#endif
            if (PROF) {
                for (int k = 0; k < 7; ++k) if (lane == (uint32_t)k) v = accW[k];
                for (int k = 0; k < 9; ++k) if (lane == 16u + k) v = accP[k];
            }
            if (a.stat_slots) {
                // one slot per wavefront; a launch of more than STAT_SLOTS wavefronts shares them (atomically)
                unsigned long long *slot = a.stat_slots + (((uint64_t)blockIdx.x * W + wave) & (STAT_SLOTS - 1)) * STAT_WORDS + lane;
                if (lane < STAT_WORDS) { if ((uint64_t)gridDim.x * W <= STAT_SLOTS) *slot += v; else atomicAdd(slot, (unsigned long long)v); }
            } else {
                if (lane < 7) atomicAdd(&a.ctl->work[lane], (unsigned long long)v);
                if (lane >= 16 && lane < 26) atomicAdd(&a.ctl->phase[lane - 16], (unsigned long long)v);
            }
        }
    }
    // ---- single-launch small call (kgpu_tokenize_batch with a handful of sentences: the reference's own call shape is
    // ONE sentence per call, src/bin/kanpyo.rs:106-126).  The grid gives every wavefront at most one sentence; input and
    // output live in pinned, device-mapped host memory; instead of three more dependent launches and four copies the
    // wavefronts meet at a counter, every one moves its own tokens to their dense place, and the last one publishes the
    // control block and the sequence number the host is polling.
    if (a.fused_host) {
        const uint32_t total_waves = gridDim.x * W;
        const uint64_t my = (uint64_t)blockIdx.x * W + wave;
        __threadfence();  // release: status, token count, staged tokens of my sentence
        if (lane == 0) atomicAdd(&a.ctl->waves_done, 1u);
        bool met = false;
        for (uint32_t spin = 0; spin < (1u << 22); ++spin) {  // bounded: every wavefront of this small grid is normally resident
            if (bcast32(__hip_atomic_load(&a.ctl->waves_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= total_waves) { met = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        // not met (the grid's workgroups were not co-resident for that long: a chip saturated by other streams): the counts read
        // below may be stale, so the call is flagged and the host redoes it on the general path instead of trusting the result
        if (!met && lane == 0) atomicExch(&a.ctl->small_abort, 1u);
        __threadfence();  // acquire
        const bool clean = met && bcast32(ld_l2(&a.ctl->ovf_count[0])) == 0 && bcast32(ld_l2(&a.ctl->small_abort)) == 0;  // every sentence was served here (else the host takes the long way)
        uint32_t before = 0, total = 0;
        if (clean) {
            for (uint64_t i0 = 0; i0 < a.n; i0 += 64) {
                const uint64_t i = i0 + lane;
                const uint32_t c = i < a.n ? ld_l2(&a.tok_count[i]) : 0u;
                before += i < my ? c : 0u;
                total += c;
            }
            before = bcast32(wave_sum(before)); total = bcast32(wave_sum(total));
            if (my < a.n) {
                const uint32_t cnt = bcast32(ld_l2(&a.tok_count[my]));
                if ((uint64_t)before + cnt <= a.out_cap) {
                    const uint32_t *src = (const uint32_t *)(a.stage + (a.offsets[my] - a.offsets[0] + my));
                    uint32_t *dst = (uint32_t *)(a.out + before);
                    for (uint32_t w = lane; w < cnt * 6; w += 64) dst[w] = ld_l2(src + w);
                }
                if (lane == 0) a.tok_offsets[my] = before;
            }
            if (my == 0 && lane == 0) a.tok_offsets[a.n] = total;
        }
        __threadfence_system();  // my part of the result is in host memory
        uint32_t copied = 0;
        if (lane == 0) copied = atomicAdd(&a.ctl->waves_copied, 1u) + 1u;
        if (bcast32(copied) == total_waves) {  // last one: publish, leave the device block zeroed for the next batch
            __threadfence();
            uint32_t *dc = (uint32_t *)a.ctl, *hc = (uint32_t *)a.fused_host;
            const uint32_t nt = (uint32_t)(offsetof(Control, n_tokens) / 4), fl = (uint32_t)(offsetof(Control, small_flag) / 4);
            for (uint32_t k = lane; k < sizeof(Control) / 4; k += 64) {
                uint32_t v = ld_l2(&dc[k]);
                if (k == nt) v = total;
                if (k == nt + 1) v = 0;
                if (k != fl) __hip_atomic_store(&hc[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dc[k] = 0;
            }
            __threadfence_system();
            if (lane == 0) __hip_atomic_store(&hc[fl], a.fused_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Workgroups of this kernel that are resident per CU for a given shape (the LDS allocation
// granularity makes this smaller than 160 KB / pool_bytes would suggest for odd sizes).
int pool_workgroups_per_cu(uint32_t pool_bytes, uint32_t waves) {
    if (pool_bytes > 64 * 1024)
        if (hipFuncSetAttribute((const void *)k_tokenize_pool<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool_bytes) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_tokenize_pool<false, false, false>, (int)(64 * waves), (size_t)pool_bytes) != hipSuccess) return 0;
    return n;
}

template <bool PROF, bool BYTE, bool WIDE>
static int launch_pool_inst(const PoolArgs &pa, uint32_t pool_bytes, uint32_t waves, int n_workgroups, void *stream) {
    if (pool_bytes > 64 * 1024) {  // beyond the default dynamic-LDS cap the kernel has to opt in
        hipError_t e = hipFuncSetAttribute((const void *)k_tokenize_pool<PROF, BYTE, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((k_tokenize_pool<PROF, BYTE, WIDE>), dim3(n_workgroups), dim3(64 * waves), pool_bytes, (hipStream_t)stream, pa);
    return (int)hipGetLastError();
}
template <bool PROF, bool BYTE>
static int launch_pool_wide(bool wide, const PoolArgs &pa, uint32_t pool_bytes, uint32_t waves, int n_workgroups, void *stream) {
    return wide ? launch_pool_inst<PROF, BYTE, true>(pa, pool_bytes, waves, n_workgroups, stream) : launch_pool_inst<PROF, BYTE, false>(pa, pool_bytes, waves, n_workgroups, stream);
}

int launch_tokenize_pool(const DictView &d, const BatchArgs &a, const WorkIO &io, uint32_t pool_bytes, uint32_t waves,
                         uint32_t max_pages, int n_workgroups, uint32_t stop_after, void *stream) {
    const uint32_t page = ((pool_bytes - POOL_HDR) / POOL_PAGES) & (waves == 1 ? ~7u : ~15u);   // (a workgroup of one wavefront owns its slice: finer pages)
    if (page == 0) return (int)hipErrorInvalidValue;
    const PoolArgs pa{d, a, io, pool_bytes, max_pages, stop_after, page, (uint32_t)((1ull << 32) / page) + 1u};
    const bool byte_walk = d.da2 == nullptr, wide = !(d.leaf_dup && d.n_unk_morph < (1u << 21));
    if (a.count_work) return byte_walk ? launch_pool_wide<true, true>(wide, pa, pool_bytes, waves, n_workgroups, stream) : launch_pool_wide<true, false>(wide, pa, pool_bytes, waves, n_workgroups, stream);
    return byte_walk ? launch_pool_wide<false, true>(wide, pa, pool_bytes, waves, n_workgroups, stream) : launch_pool_wide<false, false>(wide, pa, pool_bytes, waves, n_workgroups, stream);
}

}  // namespace kgpu

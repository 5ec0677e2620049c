// kanpyo_amd/csrc/kgpu_window.hip -- the windowed long-sentence kernel (gfx950).
//
// A sentence of any length with a BOUNDED amount of LDS: the lattice is built and relaxed window by window (WIN start
// positions at a time, ascending), every window with the phases of the LDS-resident pool kernel (kgpu_pool.hip) -- walk with
// the text and the parked matches in LDS, node-parallel emit, one parallel gather of the connection costs per block, the
// Viterbi chain at LDS latency -- and only what outlives a window leaves LDS:
//   * per node 4 + 8 bytes to HBM: its best predecessor in a dense array of its own (the backtrace reads every node's), {morph id, start char} for the
//     tokens (src/lattice.rs:144-153, src/tokenizer.rs:22-43); the byte offsets come from the per-character records of the decode pass;
//   * the bucket entries {end, dp, right id, node} of nodes that end beyond the window: those that end within the next
//     2 WIN positions stay in LDS (the carry list: they are the seeds of the following window's buckets, exactly as BOS is the
//     seed of the first, src/lattice.rs:156-164), the others -- unknown words of a long same-category run, up to 1024
//     characters ahead (src/lattice.rs:66-84) -- go to a FIFO in HBM: their end positions never decrease, so a window takes
//     what it needs from the head.  A position whose bucket is mostly such entries (the end of a 1024-character run: thousands)
//     is relaxed by streaming them from the FIFO, 64 per step.
// LDS-resident lattices (kgpu_pool.hip) do not scale to long sentences (LDS x time grows with the square of the length: DESIGN.md
// section 8); this kernel's LDS is independent of the length and its HBM traffic is one write per node.  It serves everything the pool
// kernel routes away -- from ~125 characters up to any length -- and whole batches of long sentences, which start with it (kgpu_api.cpp: ctx_pick_chain);
// for short work lists it runs as a TEAM of wavefronts per sentence (below).  (Rounds 2-3 had a second long-sentence kernel that kept the whole lattice in
// HBM and staged blocks of it in LDS for the sweep: 44 bytes per node, ~70 per byte, read back several times.  On 190-512-character
// sentences the two were level, on 2048-character documents this one is 1.6x faster: round 4 removed the other.)
//
// What it cannot hold (more than 8 + 48 dictionary prefixes in a window's overflow area, a window whose lattice outgrows the LDS budget even four
// positions long, a FIFO-order violation by a dictionary word longer than a window, node indices beyond the 16-bit window of the tie-break)
// it pushes onto the next launch's work list: the general kernel (kgpu_kernels.hip), the last resort.  Without a list to hand on to it
// flags Control::window_fail and the host reruns the batch through the general kernel.
#include "kgpu_device.h"

namespace kgpu {

using namespace dev;

namespace {

#ifndef KGPU_WIN
#define KGPU_WIN 32
#endif
constexpr uint32_t WIN = KGPU_WIN;         // start positions per window
static_assert(WIN >= 8 && WIN <= 63, "lane = position, and lane WIN holds the totals of the scans");
constexpr uint32_t REL = 2 * WIN + 1;      // bucket positions a window holds: relative ends 0 .. 2 WIN
constexpr uint32_t NEARLEN = WIN;           // a node up to this many characters long keeps its bucket entry in LDS (relative end <= 2 WIN - 1); longer ones go to the FIFO
constexpr uint32_t WMAXM = 8;              // trie matches parked per start position ...
constexpr uint32_t XMAX = 48;              // ... and this many more per window, shared (a position with more than eight dictionary prefixes parks the rest here)
constexpr uint32_t LOOKB = 192;            // text bytes staged beyond the window's own characters (a walk that runs past them reads HBM)
constexpr uint32_t TEXTB = 4 * WIN + LOOKB;
constexpr uint32_t LCODE = 96;            // character codes staged per window (char-level trie): the window's own and what the walks run into; beyond, the slab
constexpr uint32_t LTEXT = ((TEXTB + 4 > 260 ? TEXTB + 4 : 260) + 15) & ~15u;   // the staged block as aligned dwords (up to 3 bytes of misalignment in front); the decode pass's 256 + 3 bytes
static_assert(LTEXT / 4 <= 128, "two dwords per lane");
static_assert(2 * LCODE <= LTEXT && LCODE >= WIN && LCODE <= 128, "the codes share the text block's place");
constexpr uint32_t NCH_LOG = 13, NCHUNKS = 32;   // node records: chunks of 8192 (128 KB), at most 262144 nodes per sentence
constexpr uint32_t FCH_LOG = 12, FCHUNKS = 16;   // far entries: chunks of 4096 (64 KB), at most 65536 waiting at once
constexpr uint32_t WIDE_MIN = 96;          // FIFO entries at one end position from which they are streamed instead of staged in LDS
constexpr uint32_t NONE16 = 0xFFFFu;
constexpr uint32_t SLOWT = 4;               // targets relaxed together on the any-shape path (registers: a 64-bit key and a row pointer each)
constexpr uint32_t CCAP = 384;             // team mode: entries of a carry list (two banks of them in the workgroup's shared LDS)
constexpr uint32_t SEED_MARK = 0xC0000000u, SEED_FAR = 0x20000000u;   // team mode: a seed's dp is not known when its bucket slot is filled -- the slot holds
                                           // SEED_MARK | index into the carry bank (| SEED_FAR: offset into the FIFO from its head) until the sweep starts

struct Far { uint32_t end; int32_t dp; uint32_t right; uint32_t node; };          // a bucket entry that outlives its window (16 B)
struct NodeRec { int32_t sid; uint32_t start; };   // what the tokens need (8 B); the best predecessor lives in a dense array of its own (the backtrace reads every node's)

__device__ __forceinline__ uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) & ~(a - 1); }

// The double-array walk of da_walk_first (kgpu_device.h; trie/da.rs:155-182) with the text read through `byte(k)`.
template <class BY, class F>
__device__ __forceinline__ uint32_t win_walk(const DictView &d, BY &&byte, uint32_t cp, uint32_t kb, uint32_t kn, uint32_t B, int32_t base_root, F &&on_match) {
    int32_t p, bp;
    uint32_t k, nstart, steps;
    if (cp != 0xFFFFu) {
        const DaNode f = d.first[cp];
        if (f.base == 0) return (uint32_t)f.check;
        p = f.base; bp = f.check; k = kn; nstart = 1; steps = kn - kb;
    } else {  // first character outside the BMP: no table entry, walk its bytes from the root
        p = 1; bp = base_root; k = kb; nstart = 0; steps = 0;
    }
    for (;;) {
        const bool more = k < B;
        const uint32_t c = more ? byte(k) : 0u;
        const bool boundary = !more || (c & 0xC0) != 0x80;
        const uint32_t q = (uint32_t)(bp + (int32_t)c);
        const bool doprobe = boundary && nstart > 0 && (uint32_t)bp < d.da_len;
        const bool donext = more && q < d.da_len;
        DaNode t{0, 0}, nx{0, 0};
        if (doprobe) t = d.da[bp];  // + TERMINATOR (da.rs:166)
        if (donext) nx = d.da[q];
        if (doprobe && t.check == p && t.base < 0) { uint32_t id, dup; leaf_decode(d, t.base, id, dup); on_match(id, nstart, dup); }
        if (!more) break;
        ++steps;
        if (!donext || nx.check != p) break;  // da.rs:162-165
        p = (int32_t)q;
        bp = nx.base;
        nstart += boundary;
        ++k;
    }
    return steps;
}

}  // namespace

// ---- TEAM mode: TEAM wavefronts per sentence (one workgroup), for work lists shorter than the chip's wavefront slots (a lone batch of long documents fills a
// quarter of them and every document is one wavefront's dependent chain).  Window k of a sentence belongs to wavefront k mod TEAM, which runs the WHOLE window
// -- the code below, in its own region of the workgroup's LDS -- but two things chain the windows of a sentence (src/lattice.rs:101-114: the build of a position
// needs nothing of its neighbours, :116-142: the relaxation does), and each is a token passed from window to window:
//   * the STRUCTURE token: where the window starts (the one before may have been shortened to fit the LDS), the global index of its first node, which bucket
//     entries the windows before carry into it (ends, right ids, node indices -- not their costs) and which FIFO entries, the chunk tables.  Held from staging to
//     the end of emit; the next wavefront stages, walks, scans and emits its window while this one gathers its connection costs and relaxes.
//   * the VALUE token: the dp of the carried and far entries.  Held from the first relaxation to the last.
// Per window and wavefront: [structure: stage, seeds, walk, scan, emit] -> gather -> [value: seed dp, sweep, carry / far dp out] -> node records out; with TEAM = 2
// both chains (structure ~920, value ~640 of a window's ~2000 clocks per character) fit inside the other wavefront's window, so a sentence runs about twice as
// fast on twice the LDS -- which is why the host picks this only when the slots would otherwise stay empty (launch_tokenize_window).  Seeds enter a window's
// buckets before their dp exists: the slot holds a marker (SEED_MARK | index) that the value phase replaces.  The carry lists live in two banks of the
// workgroup's shared LDS (window k writes bank k & 1, window k + 1 reads it); what this form cannot hold (a carry list beyond CCAP entries) fails the
// sentence on to the next launch: the single-wavefront form of this kernel.
struct TeamState {
    uint32_t s_done, v_done;            // windows whose structure / value phase is through
    uint32_t finished, failed, why;
    uint32_t w0, gw, ncarry, fhead, ftail, last_far_end, fhead_end, wbyte0, wlim, nchunks_have, fchunks_have;
    uint32_t eos_pre;
    uint32_t go, C;                     // sentence set-up by wavefront 0: 1 = windows follow
    uint32_t slab_lo, slab_hi;
    uint32_t wT, wE;                    // work counters of the windows (PROF)
};

// PROF: device-side work counters + per-phase shader clocks (KGPU_PROFILE_WORK) -- a separate instantiation: the accumulators cost ~40 SGPRs
struct WinArgs { DictView d; BatchArgs a; WorkIO io; uint32_t lds_bytes; uint32_t claim; /* 1: the workgroups claim their sentences one by one (Control::win_ticket) */ };
#ifndef KGPU_WIN_WPE
#define KGPU_WIN_WPE 4
#endif
template <bool PROF, int TEAM>
__global__ __launch_bounds__(64 * TEAM) __attribute__((amdgpu_waves_per_eu(TEAM > 1 ? 3 : KGPU_WIN_WPE))) void k_tokenize_window(WinArgs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    // The arguments stay in the kernarg segment; every phase reads the fields it uses from there (KW_ARGS(): scalar loads behind a pointer
    // made opaque by an empty asm) -- as by-value parameters the ~90 dwords are live from entry to exit and spill (kgpu_pool.hip does the same).
    typedef const __attribute__((address_space(4))) WinArgs *KArgs;
    const KArgs kargs = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    DictView d; BatchArgs a; WorkIO io;
#define KW_ARGS() do { KArgs kq_ = kargs; asm volatile("" : "+s"(kq_)); \
        d.da = kq_->d.da; d.da_len = kq_->d.da_len; d.leaf_dup = kq_->d.leaf_dup; d.first = kq_->d.first; d.morph = kq_->d.morph; d.n_morph = kq_->d.n_morph; d.unk_morph = kq_->d.unk_morph; d.n_unk_morph = kq_->d.n_unk_morph; d.conn = kq_->d.conn; d.conn_rows = kq_->d.conn_rows; d.bos_right = kq_->d.bos_right; d.eos_left = kq_->d.eos_left; d.cat = kq_->d.cat; d.cat_len = kq_->d.cat_len; d.cinfo = kq_->d.cinfo; d.da2 = kq_->d.da2; d.da2_len = kq_->d.da2_len; d.n_nb = kq_->d.n_nb; d.crec = kq_->d.crec; d.nb_cp = kq_->d.nb_cp; d.nb_code = kq_->d.nb_code; \
        a.utf8 = kq_->a.utf8; a.offsets = kq_->a.offsets; a.n = kq_->a.n; a.ctl = kq_->a.ctl; a.arena = kq_->a.arena; a.arena_bytes = kq_->a.arena_bytes; a.stage = kq_->a.stage; a.tok_count = kq_->a.tok_count; a.status = kq_->a.status; a.count_work = kq_->a.count_work; \
        io.in_list = kq_->io.in_list; io.in_count = kq_->io.in_count; io.out_list = kq_->io.out_list; io.out_count = kq_->io.out_count; } while (0)
    KW_ARGS();
    const uint32_t lds_bytes = kargs->lds_bytes;   // of ONE wavefront's region
    const uint32_t lane = threadIdx.x & 63u, wave = TEAM > 1 ? bcast32(threadIdx.x >> 6) : 0u;
    // LDS of the workgroup: [chunk tables | TEAM > 1: team state, two carry banks] [region of wavefront 0] [region of wavefront 1] ...
    constexpr uint32_t SHARED0 = 4 * NCHUNKS + 4 * FCHUNKS;
    constexpr uint32_t SHARED = TEAM > 1 ? ((SHARED0 + (uint32_t)sizeof(TeamState) + 15u) & ~15u) + 2u * (8u * CCAP + CCAP) : SHARED0;
    TeamState *team = (TeamState *)(lds_all + SHARED0);
    uint32_t *bank_dp[2], *bank_y[2];
    uint8_t *bank_rel[2];
    if constexpr (TEAM > 1) {
        uint8_t *b0 = lds_all + ((SHARED0 + (uint32_t)sizeof(TeamState) + 15u) & ~15u);
        for (int k = 0; k < 2; ++k) { bank_dp[k] = (uint32_t *)(b0 + (size_t)k * 9u * CCAP); bank_y[k] = bank_dp[k] + CCAP; bank_rel[k] = (uint8_t *)(bank_y[k] + CCAP); }
    }
    if constexpr (TEAM > 1) if (threadIdx.x == 0) { team->nchunks_have = 0; team->fchunks_have = 0; }   // (the chunk tables persist across the workgroup's sentences; ordered by the first sentence's barrier)
    uint8_t *const lds = lds_all + SHARED + wave * lds_bytes - SHARED0;   // (the per-wavefront arrays below are carved from `lds + off0`, off0 starting behind the chunk tables' size)
    const int32_t base_root = d.da[1].base;
    Slab sa{nullptr, 0};
    uint64_t accW[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // shader clocks per phase (only summed when a.count_work): prepass, stage, seeds, walk, scan, emit, gather, sweep, flush, backtrace+tokens
#define KW_T(k) do { if constexpr (PROF) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tlast; tlast = t_; } } while (0)

    // ---- LDS, fixed part (persists across the windows of a sentence; the chunk tables across sentences) ----
    uint32_t off0 = 0;
    uint32_t *nchunk = (uint32_t *)(lds_all + off0); off0 += 4 * NCHUNKS;   // node-record chunks this workgroup owns (arena offsets / 256); shared by the team
    uint32_t *fchunk = (uint32_t *)(lds_all + off0); off0 += 4 * FCHUNKS;   // far-entry chunks
    uint8_t *ltext = lds + off0;                 off0 += LTEXT;
    uint32_t *cbw = (uint32_t *)(lds + off0);    off0 += 4 * (WIN + 2);   // byte offset of the window's characters (and one past)
    uint32_t *nb = (uint32_t *)(lds + off0);     off0 += 4 * (WIN + 2);   // nodes starting at position q -> first local node index
    uint32_t *ebase = (uint32_t *)(lds + off0);  off0 += 4 * (WIN + 2);   // first tile index per position
    uint32_t *fbase = (uint32_t *)(lds + off0);  off0 += 4 * (WIN + 2);   // first far-out slot per position
    uint32_t *boff = (uint32_t *)(lds + off0);   off0 += 4 * (REL + 2);   // bucket count -> offset, per relative end position
    uint32_t *bfill = (uint32_t *)(lds + off0);  off0 += 4 * (REL + 2);
    uint32_t *fcnt = (uint32_t *)(lds + off0);   off0 += 4 * (WIN + 2);   // FIFO entries ending at position q -> their first FIFO index
    uint32_t *wideN = (uint32_t *)(lds + off0);  off0 += 4 * (WIN + 2);   // ... how many of them are streamed (0: staged in LDS)
    uint16_t *cp16w = (uint16_t *)(lds + off0);  off0 += 2 * (WIN + 2);
    uint16_t *rlenw = (uint16_t *)(lds + off0);  off0 += 2 * (WIN + 2);
    uint16_t *uspan = (uint16_t *)(lds + off0);  off0 += 2 * (WIN + 2);
    uint8_t *catw = lds + off0;                  off0 += align_up(WIN + 2, 4);
    uint8_t *mcnt = lds + off0;                  off0 += align_up(WIN + 2, 4);   // dictionary prefixes at the position (all of them: beyond WMAXM in xbuf)
    uint32_t *xcnt = (uint32_t *)(lds + off0);   off0 += 4;
    uint2 *xbuf = (uint2 *)(lds + align_up(off0, 8)); off0 = align_up(off0, 8) + 8 * XMAX;   // {position | chars << 8, id | records << 21 ... as parked}: {pos | nch << 8 | nrec << 16, id}
    off0 = align_up(off0, 16);
    uint32_t nchunks_have = 0, fchunks_have = 0;   // wave-uniform; the tables persist in LDS
    const uint32_t MS = (d.leaf_dup && d.n_unk_morph < (1u << 21)) ? 4u : 8u;
    const uint32_t mbytes = WIN * WMAXM * MS;
    const bool cfg_bad = off0 + mbytes + 2048 > lds_bytes;  // (launch configuration error: every sentence is handed back)
    // The carry list lives at the top of the LDS, just below the match buffer: n entries {dp, right id | node index relative to the NEXT
    // window's base} + their relative end positions.  It is written when a window is flushed (the pair table that occupied the region
    // is dead by then) and read when the next window's buckets are seeded.
    const uint32_t moff = (lds_bytes - mbytes) & ~15u;
    auto carry8 = [&](uint32_t n) { return (uint2 *)(lds + moff - align_up(8 * n, 16)); };
    auto crel = [&](uint32_t n) { return lds + moff - align_up(8 * n, 16) - align_up(n, 16); };
    auto cbytes = [&](uint32_t n) { return TEAM > 1 ? 0u : align_up(8 * n, 16) + align_up(n, 16); };   // (team mode: the carry lists live in the shared banks)
    // team tokens: LDS words, polled (the wavefronts of a team never meet at a barrier inside a sentence's windows)
    auto lds_get = [&](uint32_t *w) { return bcast32(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    auto lds_put = [&](uint32_t *w, uint32_t v) { if (lane == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto team_release = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); };
    auto team_acquire = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };

    auto fail = [&](uint64_t s) {  // this sentence needs the HBM-lattice kernel: on to the next launch's list, or (no list) the host reruns the batch
        if (io.out_list) { work_defer(io, lane, s); if (lane == 0) a.tok_count[s] = 0; }
        else if (lane == 0) { a.status[s] = KGPU_SENT_NO_SCRATCH; a.tok_count[s] = 0; atomicExch(&a.ctl->window_fail, 1u); }
    };
    auto chunk_get = [&](uint32_t *table, uint32_t &have, uint32_t idx, uint32_t bytes) -> bool {  // make chunk idx exist (bytes: a multiple of 256)
        while (have <= idx) {
            uint64_t o = 0;
            if (lane == 0) o = atomicAdd(&a.ctl->arena_cursor, (unsigned long long)bytes);
            o = bcast64(o);
            if (o + bytes > a.arena_bytes || (o >> 8) > 0xFFFFFFFFull) { if (lane == 0) atomicExch(&a.ctl->arena_overflow, 1u); return false; }
            if (lane == 0) table[have] = (uint32_t)(o >> 8);
            ++have;
            wave_sync();
        }
        return true;
    };
    // a node chunk: [pre: u32 x 8192 | NodeRec x 8192] = 96 KB
    auto node_pre = [&](uint32_t g) { return (uint32_t *)(a.arena + ((uint64_t)nchunk[g >> NCH_LOG] << 8)) + (g & ((1u << NCH_LOG) - 1)); };
    auto node_rec = [&](uint32_t g) { return (NodeRec *)(a.arena + ((uint64_t)nchunk[g >> NCH_LOG] << 8) + (4u << NCH_LOG)) + (g & ((1u << NCH_LOG) - 1)); };
    auto far_rec = [&](uint32_t f) { return (Far *)(a.arena + ((uint64_t)fchunk[(f >> FCH_LOG) % FCHUNKS] << 8)) + (f & ((1u << FCH_LOG) - 1)); };

    for (uint32_t iter = 0;; ++iter) {
        KW_ARGS();
        uint64_t s = 0;
        if (TEAM == 1 && kargs->claim) {
            // the ordinary form over a list longer than its grid claims its sentences one by one (one atomic per sentence of a kernel that spends 100 us and more on
            // each): the launch -- cfg 3 in batches of 65 536: five long sentences per workgroup -- no longer ends with the workgroup whose every-G-th share was the
            // longest (25.5 -> 27.5 M sentences/s).  The host asks for it by the list's expected length: with a sentence per workgroup the static form is the better one
            // (cfg 3 at 4096 per batch 21.8 against 21.2).
            uint32_t tk = 0;
            if (lane == 0) tk = atomicAdd(&a.ctl->win_ticket, 1u);
            tk = bcast32(tk);
            if (!work_next_at(io, a, tk, s)) break;
        } else if (!work_next(io, a, iter, s)) break;
        const uint64_t b0 = a.offsets[s];
        const uint32_t B = (uint32_t)(a.offsets[s + 1] - b0);
        const uint8_t *text = a.utf8 + b0;
        uint64_t tlast = PROF ? __builtin_amdgcn_s_memtime() : 0;
        // ---- sentence set-up and the two prepasses: the workgroup's first wavefront (a team's others wait at the barrier below)
        uint2 *crec = nullptr; uint32_t *path = nullptr; uint16_t *code16 = nullptr;
        uint32_t C = 0;
        const uint64_t na = (uint64_t)B + 4;
        const bool ct = d.da2 != nullptr;
        auto setup_fence = [&]() { if constexpr (TEAM == 1) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } };
        if constexpr (TEAM > 1) __syncthreads();   // the sentence before is through on every wavefront (its backtrace read what the others wrote; the slab is free)
        bool go = true;
        if (TEAM == 1 || wave == 0) go = [&]() -> bool {
            if (cfg_bad) { fail(s); return false; }

        // ---- slab: one 8-byte record per character in HBM (written by the two prepasses, read once per window), then the backtrace's path ----
        //   .x = byte offset (24 bits) | category << 24     .y = BMP code point (0xFFFF: not BMP) | same-category run length from here << 16
            if (B >= (1u << 24)) { fail(s); return false; }
        if (!slab_ensure(sa, na * 14 + 64, a, lane)) {
            if (lane == 0) { a.status[s] = KGPU_SENT_NO_SCRATCH; a.tok_count[s] = 0; }
            return false;
        }
        crec = (uint2 *)sa.ptr;              // [C]: {B, 0}
        path = (uint32_t *)(crec + na);      // backtrace
        code16 = (uint16_t *)(path + na);    // char-level trie: the characters' codes (0xFFFF: in no key), [C] = 0xFFFF

        // ---- pass 0: decode + validate + category (char_category_def.rs:33-38), 256 bytes a round: the round's text goes through LDS (the
        // continuation bytes are read there), the next round's is in flight meanwhile, and the four category loads of a round are issued together --
        // one memory round trip per 256 bytes (a 64-byte round with its dependent category load made this pass 13 % of a 2048-character document)
#ifdef KGPU_WIN_SPLIT  // measurement build: slot 0 = sentence set-up, 1 (+ stage) = decode pass, 2 (+ seeds) = run-length pass
        KW_T(0);
#endif
        uint32_t bad = 0, lensum = 0;
        {
            uint32_t pf[5];
            auto fetch = [&](uint32_t k0) {  // (unconditional loads at clamped addresses: a load under a lane mask is waited for on the spot)
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const uint32_t k = k0 + 64u * (uint32_t)u + (u < 4 ? lane : (lane & 3u));
                    pf[u] = text[min(k, B - 1u)];
                }
            };
            if (B) fetch(0);
            for (uint32_t k0 = 0; k0 < B; k0 += 256) {
                wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) ltext[64 * u + lane] = (uint8_t)(k0 + 64u * (uint32_t)u + lane < B ? pf[u] : 0x80u);
                if (lane < 3) ltext[256 + lane] = (uint8_t)(k0 + 256u + lane < B ? pf[4] : 0x80u);
                wave_sync();
                fetch(k0 + 256);  // (clamped addresses: harmless past the end)
                uint32_t ci[4], kk[4], cpx[4];
                bool st[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t r = 64u * (uint32_t)u + lane, k = k0 + r;
                    const uint32_t b = ltext[r];
                    const bool start = k < B && (b & 0xC0) != 0x80;
                    const uint64_t m = __ballot(start);
                    ci[u] = C + __popcll(m & ((1ull << lane) - 1));
                    st[u] = start; kk[u] = k; cpx[u] = 0;
                    if (start) {
                        uint32_t cp;
                        const uint32_t l = utf8_decode_lead(b, k, B, [&](uint32_t kk) -> uint32_t { return ltext[kk - k0]; }, cp, bad);
                        lensum += l;
                        cpx[u] = cp;
                    }
                    C += __popcll(m);
                }
                uint32_t cv[4], cd[4];
                if (ct) {   // category and code from the character's record (kgpu_chartrie.cpp); what the table cannot name: the slow way
                    CharRec rr[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) rr[u] = d.crec[cpx[u] < 0xFFFFu ? cpx[u] : 0u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cv[u] = rr[u].cat; cd[u] = rr[u].code;
                        if (st[u] && cpx[u] >= 0xFFFFu) { cv[u] = d.cat[cpx[u] < d.cat_len ? cpx[u] : 0u]; cd[u] = d.n_nb ? ct_code_nonbmp(d, cpx[u]) : 0xFFFFu; }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { cv[u] = d.cat[cpx[u] < d.cat_len ? cpx[u] : 0u]; cd[u] = cpx[u] < 0xFFFFu ? cpx[u] : 0xFFFFu; }  // (cpx = 0 where no character starts: four loads, one wait)
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (st[u]) {
                        crec[ci[u]].x = kk[u] | ((bad ? 0u : cv[u]) << 24);
                        *(uint16_t *)&crec[ci[u]].y = (uint16_t)cd[u];   // the code point (byte-level walk), or the code
                        if (ct) code16[ci[u]] = (uint16_t)cd[u];
                    }
            }
        }
        lensum = bcast32(wave_sum(lensum));
        if (__ballot(bad != 0) != 0 || lensum != B) {
            if (lane == 0) { a.status[s] = KGPU_SENT_INVALID_UTF8; a.tok_count[s] = 0; }
            return false;
        }
        if (lane == 0) { crec[C] = make_uint2(B, 0u); if (ct) code16[C] = 0xFFFFu; }
        setup_fence();
#ifdef KGPU_WIN_SPLIT
        KW_T(1);
#endif
        KW_ARGS();
        // ---- pass 1 (descending): length of the same-category run that starts at each character, capped at 1024; 256 characters a round ----
        {
            uint32_t carry_end = C;
            for (int ch = (int)((C + 255) / 256) - 1; ch >= 0; --ch) {
                uint32_t cat[4], ncat[4];
#pragma unroll
                for (int u = 3; u >= 0; --u) {
                    const uint32_t i = (uint32_t)ch * 256 + 64u * (uint32_t)u + lane;
                    cat[u] = crec[min(i, C)].x >> 24;   // ([C] exists: loads without a lane mask, eight in flight)
                    ncat[u] = crec[min(i + 1, C)].x >> 24;
                }
#pragma unroll
                for (int u = 3; u >= 0; --u) {
                    const uint32_t i = (uint32_t)ch * 256 + 64u * (uint32_t)u + lane;
                    const bool active = i < C;
                    const uint64_t bm = __ballot(active && (i + 1 >= C || ncat[u] != cat[u]));
                    const uint64_t rest = bm >> lane;
                    const uint32_t run_end = rest ? i + (uint32_t)__ffsll((unsigned long long)rest) : carry_end;
                    carry_end = bcast32(run_end);
                    if (active) { const uint32_t r = run_end - i; ((uint16_t *)&crec[i].y)[1] = (uint16_t)(r < MAX_UNKNOWN_LEN ? r : MAX_UNKNOWN_LEN); }
                }
            }
        }
        setup_fence();

#ifdef KGPU_WIN_SPLIT
        KW_T(2);
#else
        KW_T(0);
#endif
        return true;
        }();
        if constexpr (TEAM > 1) {
            if (wave == 0 && lane == 0) {
                team->go = go ? 1u : 0u; team->C = C; team->slab_lo = (uint32_t)(uintptr_t)sa.ptr; team->slab_hi = (uint32_t)((uintptr_t)sa.ptr >> 32);
                team->s_done = 0; team->v_done = 0; team->finished = 0; team->failed = 0; team->why = 0; team->eos_pre = NONE; team->wT = 0; team->wE = 0;
                team->w0 = 0; team->gw = 1; team->ncarry = 1; team->fhead = 0; team->ftail = 0; team->last_far_end = 0; team->fhead_end = 0xFFFFFFFFu; team->wbyte0 = 0; team->wlim = WIN;
                bank_dp[1][0] = 0u; bank_y[1][0] = d.bos_right << 1; bank_rel[1][0] = 0;   // BOS: node 0, ends at 0, dp None -> 0 (lattice.rs:127,156-164): what "window -1" carries
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            go = bcast32(team->go) != 0;
            if (go && wave != 0) {
                C = bcast32(team->C);
                crec = (uint2 *)(((uintptr_t)bcast32(team->slab_hi) << 32) | (uintptr_t)bcast32(team->slab_lo));
                path = (uint32_t *)(crec + na);
                code16 = (uint16_t *)(path + na);
            }
        }
        if (!go) continue;
        // ---- the windows ----
        uint32_t w0 = 0, gw = 1 /* global index of the window's first node: BOS is node 0 */, ncarry = 1, fhead = 0, ftail = 0, last_far_end = 0;
        uint32_t fhead_end = 0xFFFFFFFFu;   // end position of the FIFO's head entry (0xFFFFFFFF: the FIFO is empty)
        uint32_t wT = 0, wE = 0;
        bool failed = false;
        uint32_t why = 0;  // which limit a failed sentence ran into (Control::phase[why] counts them: KGPU_WINDOW_TRACE)
        if constexpr (TEAM == 1) if (lane == 0) { carry8(1)[0] = make_uint2(0u, d.bos_right << 1); crel(1)[0] = 0; }  // BOS: node 0, ends at 0, dp None -> 0 (lattice.rs:127,156-164)
        uint32_t wbyte0 = 0;  // first byte of the next window's characters
        uint2 pf_rec = make_uint2(0u, 0u);
        uint32_t pf_t0 = 0, pf_t1 = 0;
        bool staged = false;
        auto prefetch = [&](uint32_t w0n, uint32_t tbn) {  // the records of positions w0n .. w0n + WIN and the text block from byte tbn, as aligned dwords
            if (lane <= WIN && w0n + lane <= C) pf_rec = crec[w0n + lane];
            if (ct) {   // the codes of positions w0n .. w0n + LCODE - 1 (past the end: [C], "none")
                pf_t0 = code16[min(w0n + lane, C)];
                if (lane < LCODE - 64) pf_t1 = code16[min(w0n + 64 + lane, C)];
                return;
            }
            const uintptr_t g = (uintptr_t)(text + tbn);
            const uint32_t mis = (uint32_t)(g & 3u), tl = min(B - tbn, TEXTB) + mis;
            const uint32_t *g32 = (const uint32_t *)(g - mis);
            if (4 * lane < tl) pf_t0 = g32[lane];
            if (4 * (64 + lane) < tl) pf_t1 = g32[64 + lane];
        };
        if constexpr (TEAM == 1) prefetch(0, 0);
        uint32_t wlim = WIN;  // positions per window: halved when a window's lattice outgrows the LDS, grown again by how empty the LDS was (see the end of the loop)
        wave_sync();
        uint32_t eos_pre = NONE;
        uint32_t kwin = wave;                     // team: the index of this wavefront's next window
        bool holding = false, have_v = false;     // team: inside the structure phase (a window that is redone shorter keeps the token) / the value phase
        for (; TEAM > 1 || (w0 <= C && !failed); ) {
            if constexpr (TEAM > 1) {
                if (!holding) {   // the structure token: window kwin - 1 has been emitted (or the sentence is over, one way or the other)
                    bool stop = false;
                    for (;;) {
                        if (lds_get(&team->failed) | lds_get(&team->finished)) { stop = true; break; }
                        if (lds_get(&team->s_done) == kwin) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (stop) break;
                    team_acquire();
                    w0 = bcast32(team->w0); gw = bcast32(team->gw); ncarry = bcast32(team->ncarry); fhead = bcast32(team->fhead); ftail = bcast32(team->ftail);
                    last_far_end = bcast32(team->last_far_end); fhead_end = bcast32(team->fhead_end); wbyte0 = bcast32(team->wbyte0); wlim = bcast32(team->wlim);
                    nchunks_have = bcast32(team->nchunks_have); fchunks_have = bcast32(team->fchunks_have);
                    if (w0 > C) { lds_put(&team->finished, 1u); break; }   // the window before was the last one
                    holding = true; have_v = false; staged = false;
                    prefetch(w0, wbyte0);   // (nobody could request this window's records ahead: where it starts was not known)
                }
            }
            const uint32_t pb = (kwin + 1u) & 1u;   // team: the carry bank the window before wrote
            const uint32_t nw = min(wlim, C + 1 - w0);         // positions of this window; position C (if in it) holds only EOS
            const uint32_t nwc = min(nw, C - w0);              // ... of which characters
            const uint32_t rb = gw >= 0x8000u ? gw - 0x8000u : 0u;   // node indices inside the window's LDS are 16-bit offsets from here
            // -- stage: the window's per-character records and its text (+ LOOKB bytes) into LDS -- from the registers they were prefetched into
            // while the previous window was relaxed (the text block starts at the byte the previous window's characters ended at and has a
            // fixed length).  A window that is redone half as long finds everything still in place.
            const uint32_t tb0 = wbyte0, tlen = min(B - tb0, TEXTB);
            const uint32_t tmis = (uint32_t)((uintptr_t)(text + tb0) & 3u);
            if (!staged) {
                const uint32_t nfull = min(WIN, C - w0);
                if (lane <= nfull) cbw[lane] = pf_rec.x & 0xFFFFFFu;
                if (lane < nfull) { cp16w[lane] = (uint16_t)pf_rec.y; catw[lane] = (uint8_t)(pf_rec.x >> 24); rlenw[lane] = (uint16_t)(pf_rec.y >> 16); }
                if (ct) {
                    ((uint16_t *)ltext)[lane] = (uint16_t)pf_t0;
                    if (lane < LCODE - 64) ((uint16_t *)ltext)[64 + lane] = (uint16_t)pf_t1;
                } else {
                    ((uint32_t *)ltext)[lane] = pf_t0;
                    if (lane < LTEXT / 4 - 64) ((uint32_t *)ltext)[64 + lane] = pf_t1;
                }
                staged = true;
            }
            for (uint32_t e = lane; e < REL + 2; e += 64) { boff[e] = 0; bfill[e] = 0; }
            if (lane < WIN + 2) { fcnt[lane] = 0; wideN[lane] = 0; }
            if (lane == 0) *xcnt = 0;
            wave_sync();
            const uint32_t wbyte_next = cbw[nwc];
            auto byte = [&](uint32_t k) -> uint32_t { const uint32_t r = k - tb0; return r < tlen ? ltext[r + tmis] : text[k]; };

            KW_T(1);
            KW_ARGS();
            // -- seeds: carried entries per relative end; FIFO entries that end inside this window
            uint32_t seed_bad = 0;
            for (uint32_t k = lane; k < ncarry; k += 64) atomicAdd(&boff[TEAM > 1 ? bank_rel[pb][k] : crel(ncarry)[k]], 1u);
            uint32_t fin = 0;  // FIFO entries [fhead, fhead + fin) end inside this window
            uint32_t next_head_end = fhead_end;   // the end position of the entry that will head the FIFO once this window is through
            // (the head's end position is known from the scan that stopped at it: while it lies beyond the window no entry ends inside it -- ends never
            // decrease -- and the FIFO is not read at all: through a 1024-character run that is one round trip less for every one of its 32 windows)
            if (fhead < ftail && fhead_end < w0 + nw) {
                next_head_end = 0xFFFFFFFFu;
                for (uint32_t f0 = fhead; f0 < ftail; f0 += 64) {
                    const uint32_t f = f0 + lane;
                    bool in = false;
                    uint32_t e_end = 0xFFFFFFFFu;
                    if (f < ftail) { const Far e = *far_rec(f); e_end = e.end; in = e.end < w0 + nw; if (in) { atomicAdd(&fcnt[e.end - w0], 1u); if (e.node < rb) seed_bad = 1; } }
                    const uint64_t m = __ballot(in);
                    fin += __popcll(m);
                    if (m != ~0ull) {  // the first entry beyond the window ends the scan (past the tail: the FIFO is exhausted, "none")
                        next_head_end = (uint32_t)__builtin_amdgcn_readlane((int)e_end, (int)(__ffsll((unsigned long long)~m) - 1));
                        break;
                    }
                }
            }
            fin = bcast32(fin);
            if (__ballot(seed_bad != 0) != 0) { failed = true; why = 1; break; }
            wave_sync();
            {   // FIFO entries per position -> first FIFO index; many at one position: streamed (wide), else staged as seeds
                const uint32_t v = lane < nw ? fcnt[lane] : 0u;
                const uint32_t vs = wave_incl_scan(v, lane);
                if (lane < nw) {
                    fcnt[lane] = fhead + vs - v;
                    const bool wide = v >= WIDE_MIN;
                    wideN[lane] = wide ? v : 0u;
                    if (!wide && v) atomicAdd(&boff[lane], v);
                }
            }
            wave_sync();

            KW_T(2);
            KW_ARGS();
            // -- walk: one double-array walk per start position; count, park the matches (lattice.rs:24-38, 42-99)
            uint32_t *mbuf = (uint32_t *)(lds + moff);
            uint32_t ovf = 0, cnt = 0, nfar = 0, wTw = 0, wEw = 0;  // (work of this window: added to the sentence's only when the window goes through)
            if (lane < nwc) {
                uint32_t m = 0;
                auto on_match = [&](uint32_t id, uint32_t nch, uint32_t dup) {
                    const uint32_t nrec = 1u + (dup != NONE ? dup : (uint32_t)d.morph[id - 1].dup);  // index.rs:46-51
                    if (m < WMAXM && nch < 256) {
                        if (MS == 4) mbuf[lane * WMAXM + m] = id | (nch << 21) | ((nrec < 8 ? nrec : 0u) << 29);
                        else *(uint2 *)(mbuf + 2 * (lane * WMAXM + m)) = make_uint2(id, nch | (nrec << 8));
                    } else if (nch < 256 && m < 255 && nrec < 65536) {   // the ninth prefix and beyond: the window's shared overflow area (a lane's entries keep their order)
                        const uint32_t k = atomicAdd(xcnt, 1u);
                        if (k < XMAX) xbuf[k] = make_uint2(lane | (nch << 8) | (nrec << 16), id); else ovf = 1;
                    } else ovf = 1;
                    ++m;
                    cnt += nrec;
                    if (nch <= NEARLEN) atomicAdd(&boff[lane + nch], nrec); else nfar += nrec;
                };
                if (ct) {
                    const uint16_t *lcode = (const uint16_t *)ltext;
                    ct_walk(d, 1, d.da2[1].base, 0, [&](uint32_t dep) -> uint32_t { const uint32_t j = lane + dep; return j < LCODE ? (uint32_t)lcode[j] : (uint32_t)code16[min(w0 + j, C)]; }, on_match, 0u);
                    if constexpr (PROF) if (!(a.count_work & 2u)) wTw += da_walk(d, text, cbw[lane], B, base_root, [](uint32_t, uint32_t, uint32_t) {});  // the reference's byte steps (work counters; not in phase-timing runs)
                } else
                wTw += win_walk(d, byte, cp16w[lane], cbw[lane], cbw[lane + 1], B, base_root, on_match);
                mcnt[lane] = (uint8_t)m;
                const CatInfo ci = d.cinfo[catw[lane]];
                uint32_t span = 0;
                if ((cnt == 0 || (ci.flags & CAT_INVOKE)) && (ci.flags & CAT_HAS_UNK) && ci.unk_count) {  // lattice.rs:54,87-92
                    span = (ci.flags & CAT_GROUP) ? (uint32_t)rlenw[lane] : 1u;                             // lattice.rs:66-84
                    cnt += ci.unk_count;
                    if (span <= NEARLEN) atomicAdd(&boff[lane + span], ci.unk_count); else nfar += ci.unk_count;
                    if (m < WMAXM) {
                        if (MS == 4) mbuf[lane * WMAXM + m] = (uint32_t)ci.unk_first | ((ci.unk_count < 8 ? ci.unk_count : 0u) << 29);
                        else *(uint2 *)(mbuf + 2 * (lane * WMAXM + m)) = make_uint2((uint32_t)ci.unk_first, ci.unk_count);
                    }
                }
                uspan[lane] = (uint16_t)span;
            } else if (lane < nw) { cnt = 1; mcnt[lane] = 0; uspan[lane] = 0; }  // position C: EOS (lattice.rs:165-175)
            if (__ballot(ovf != 0) != 0) { failed = true; why = 2; break; }
            wave_sync();

            KW_T(3);
            KW_ARGS();
            // -- scan: node numbering (insertion order, lattice.rs:105-110), bucket offsets, tile offsets, far-out slots
            uint32_t N, Nb, NT, NF;
            {
                const uint32_t v = lane < nw ? cnt : 0u, fo = lane < nw ? nfar : 0u;
                const uint32_t vs = wave_incl_scan(v, lane), fs = wave_incl_scan(fo, lane);
                if (lane <= nw) { nb[lane] = vs - v; fbase[lane] = fs - fo; }
                N = (uint32_t)__builtin_amdgcn_readlane((int)vs, 63); NF = (uint32_t)__builtin_amdgcn_readlane((int)fs, 63);
                // bucket offsets over the REL relative ends (two rounds of 64)
                uint32_t bc = 0;
                for (uint32_t e0 = 0; e0 < REL + 1; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    const uint32_t w = e < REL ? boff[e] : 0u;
                    const uint32_t ws = wave_incl_scan(w, lane);
                    if (e <= REL) boff[e] = bc + ws - w;
                    bc += (uint32_t)__builtin_amdgcn_readlane((int)ws, 63);
                }
                Nb = bc;
                wave_sync();
                // stage B: tiles (kgpu_device.h) for every position whose predecessors are all in LDS; a position with STREAMED predecessors (wideN: the end of a
                // long same-category run -- six unknown words per start position end there, lattice.rs:66-84) keeps the any-shape step, which loads its costs itself;
                // a position nothing ends at has no step at all
                const uint32_t P = lane < nw ? boff[lane + 1] - boff[lane] : 0u;
                const bool tiled = lane < nw && wideN[lane] == 0 && v != 0 && P != 0;
                const uint32_t x = tiled ? ((v + 7u) >> 3) * ((P + 7u) >> 3) : 0u;
                const uint32_t xs = wave_incl_scan(x, lane);
                if (lane <= nw) ebase[lane] = xs - x;
                NT = (uint32_t)__builtin_amdgcn_readlane((int)xs, 63);
                wEw = (lane < nw) ? v * (P + wideN[lane]) : 0u;   // relaxations: targets x (LDS predecessors + streamed ones)
            }
            // -- LDS carve: buckets (+ far-out slots + sink), node arrays (+ the padding tiles' target), far-out ends, the tile list (overlays the match buffer)
            uint32_t off = off0;
            uint2 *bk = (uint2 *)(lds + off);            off += 8 * (Nb + NF + 1);
            uint8_t *brel = lds + off;                   off += align_up(Nb, 8);   // relative end position of every bucket slot (the flush carries slot by slot)
            uint2 *node = (uint2 *)(lds + off);          off += 8 * (N + 1);       // {word cost | bucket slot << 16, row offset of the left id -> (low half) best predecessor}: kgpu_device.h
            int32_t *nSid = (int32_t *)(lds + off);      off += 4 * N;
            uint32_t *farEnd = (uint32_t *)(lds + off);  off += 4 * NF;
            uint16_t *nStart = (uint16_t *)(lds + off);  off += 2 * N;
            off = align_up(off, 8);
            uint2 *tiles = (uint2 *)(lds + off);
            const uint32_t pair_need = 8 * NT;
            // What this window needs of the LDS (the three layouts that must fit: emit, sweep, flush), and from it the length that WOULD fill 7/8 of the
            // budget at this lattice density: the part above the fixed arrays grows with the positions.  A window that does not fit is redone that long
            // (it has been walked and counted for nothing -- round 3 halved it, and doubled back after every success: through a dense stretch every
            // other window was thrown away); the next window starts that long as well.
            const uint32_t lds_used = max(max(off + cbytes(ncarry) + mbytes + 16, off + pair_need),
                                          off0 + 8 * (Nb + NF + 1) + align_up(Nb, 8) + cbytes(Nb - boff[nw]) + (lds_bytes - moff));
            const uint32_t lds_base = off0 + mbytes + 16;
            const uint32_t fit_len = (nw * (lds_bytes - lds_base) * 7u) / (max(lds_used, lds_base + 1u) - lds_base) / 8u;   // (63 x 160 KB x 7 < 2^32)
            if (N > 0x7FFF || NF >= 0x7FFF || Nb + NF + 1 > SLOT_MAX || lds_used > lds_bytes) {   // (SLOT_MAX: a node carries its slot's byte offset in a half-word)
                if (wlim > 4 && nw > 1) { wlim = max(4u, min(nw - 1, fit_len)); continue; }  // the same window once more, as long as the LDS allows at this density
                failed = true; why = 8;
                break;
            }
            if ((uint64_t)gw + N >= ((uint64_t)NCHUNKS << NCH_LOG)) { failed = true; why = 3; break; }
            if (TEAM > 1 && Nb - boff[nw] > CCAP) { failed = true; why = 7; break; }   // the carry list does not fit a bank: the single-wavefront form takes the sentence
            if constexpr (TEAM == 1) if (w0 + nw <= C) prefetch(w0 + nw, wbyte_next);  // the next window's stage: in flight while this one is emitted and relaxed
            wave_sync();

            KW_T(4);
            KW_ARGS();
            // -- emit 3a (lane = start position, LDS only): the node list in insertion order; node[].y = relative end, or 0x8000 | far-out slot
            if (lane < nwc) {
                uint32_t t = nb[lane], fslot = fbase[lane];
                const uint32_t nm_all = mcnt[lane], nm = min(nm_all, WMAXM), span = uspan[lane];
                uint32_t ufirst = 0, ucnt = 0;
                if (span) {
                    if (nm < WMAXM) {
                        if (MS == 4) { const uint32_t w = mbuf[lane * WMAXM + nm]; ufirst = w & 0x1FFFFFu; ucnt = w >> 29; }
                        else { const uint2 w = *(const uint2 *)(mbuf + 2 * (lane * WMAXM + nm)); ufirst = w.x; ucnt = w.y; }
                    }
                    if (ucnt == 0) { const CatInfo ci = d.cinfo[catw[lane]]; ufirst = (uint32_t)ci.unk_first; ucnt = ci.unk_count; }
                }
                auto put = [&](int32_t sid, uint32_t rel) {
                    nSid[t] = sid; nStart[t] = (uint16_t)lane;
                    if (rel - lane <= NEARLEN) node[t].y = rel;
                    else { node[t].y = 0x8000u | fslot; farEnd[fslot] = w0 + rel; ++fslot; }
                    ++t;
                };
                for (uint32_t m = 0; m < nm; ++m) {
                    uint32_t id, nch, nrec;
                    if (MS == 4) {
                        const uint32_t w = mbuf[lane * WMAXM + m];
                        id = w & 0x1FFFFFu; nch = (w >> 21) & 255u; nrec = w >> 29;
                        if (nrec == 0) nrec = 1u + d.morph[id - 1].dup;
                    } else {
                        const uint2 w = *(const uint2 *)(mbuf + 2 * (lane * WMAXM + m));
                        id = w.x; nch = w.y & 255u; nrec = w.y >> 8;
                    }
                    for (uint32_t r = 0; r < nrec; ++r) put((int32_t)(id + r), lane + nch);
                }
                if (nm_all > WMAXM) {   // the longer prefixes, from the overflow area, in the order the walk found them (ascending length: trie/da.rs:155-182)
                    const uint32_t nx = min(*xcnt, XMAX);
                    for (uint32_t k = 0; k < nx; ++k) {
                        const uint2 w = xbuf[k];
                        if ((w.x & 255u) != lane) continue;
                        const uint32_t nch = (w.x >> 8) & 255u, nrec = w.x >> 16;
                        for (uint32_t r = 0; r < nrec; ++r) put((int32_t)(w.y + r), lane + nch);
                    }
                }
                if (span) for (uint32_t r = 0; r < ucnt; ++r) put(-(int32_t)(ufirst + r), lane + span);
            }
            if (nw > nwc && lane == nwc) {  // EOS: Morph(0,0,0), its dp goes to the sink slot
                const uint32_t t = nb[lane];
                nSid[t] = 0; nStart[t] = (uint16_t)lane; node[t].y = 0x7FFFu;
            }
            wave_sync();
            // -- seeds into their buckets: carried entries, then the staged FIFO entries
            for (uint32_t k = lane; k < ncarry; k += 64) {
                const uint32_t rel = TEAM > 1 ? (uint32_t)bank_rel[pb][k] : (uint32_t)crel(ncarry)[k];
                const uint32_t slot = boff[rel] + atomicAdd(&bfill[rel], 1u);
                if constexpr (TEAM > 1) bk[slot] = make_uint2(SEED_MARK | k, bank_y[pb][k]);   // (its dp follows with the value token)
                else bk[slot] = carry8(ncarry)[k];  // (its node index is already relative to this window's base)
                brel[slot] = (uint8_t)rel;
            }
            for (uint32_t f = fhead + lane; f < fhead + fin; f += 64) {
                const Far e = *far_rec(f);
                const uint32_t rel = e.end - w0;
                if ((wideN[rel] & 0x7FFFFFFFu) == 0) {
                    const uint32_t slot = boff[rel] + atomicAdd(&bfill[rel], 1u);
                    bk[slot] = make_uint2(TEAM > 1 ? (SEED_MARK | SEED_FAR | (f - fhead)) : (uint32_t)e.dp, (e.right & 0xFFFFu) | ((e.node - rb) << 16));
                    brel[slot] = (uint8_t)rel;
                }
            }
            // -- emit 3b (lane = node): morph record, bucket slot
            const uint32_t rows2 = d.conn_rows * 2u;   // bytes per matrix row (connection.rs:12-14)
            for (uint32_t t0 = 0; t0 < N; t0 += 256) {
                uint32_t tt[4], ee[4];
                Morph8 mm[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tt[k] = t0 + 64 * k + lane;
                    const bool v = tt[k] < N;
                    const int32_t sid = v ? nSid[tt[k]] : 1;
                    ee[k] = v ? node[tt[k]].y : 0u;
                    mm[k] = sid == 0 ? Morph8{(int16_t)d.eos_left, 0, 0, 0} : d.morph[sid > 0 ? (uint32_t)sid - 1u : d.n_morph - 1u - (uint32_t)sid];   // (one table: DictView)
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (tt[k] < N) {
                        uint32_t slot;
                        if (ee[k] == 0x7FFFu) slot = Nb + NF;                                   // EOS: the sink
                        else if (ee[k] & 0x8000u) slot = Nb + (ee[k] & 0x7FFFu);                 // ends beyond the LDS buckets: a far-out slot
                        else { slot = boff[ee[k]] + atomicAdd(&bfill[ee[k]], 1u); brel[slot] = (uint8_t)ee[k]; }
                        node[tt[k]] = make_uint2((uint32_t)(uint16_t)mm[k].cost | (slot << SLOT_SHIFT), (uint32_t)(uint16_t)mm[k].left * rows2);
                        bk[slot] = make_uint2((uint32_t)INF, ((uint32_t)(uint16_t)mm[k].right << 1) | ((gw + tt[k] - rb) << 16));   // (2 * right: ids are non-negative i16)
                    }
                }
            }
            wave_sync();

            [[maybe_unused]] const uint32_t fhead_w = fhead, ftail_w = ftail;   // the FIFO as this window found it
            if constexpr (TEAM > 1) {
                // ---- the structure of this window is complete: hand the structure token on.  Everything the next window needs that is not a dp -- the node
                // chunks, the far entries' ends / right ids / nodes in the FIFO, the carried entries' right ids / nodes / relative ends in this window's bank,
                // where the next window starts -- is written here; the dp of both follow with the value token.
                KW_ARGS();
                if (!chunk_get(nchunk, nchunks_have, (gw + N - 1) >> NCH_LOG, (4u + (uint32_t)sizeof(NodeRec)) << NCH_LOG)) { failed = true; why = 3; break; }
                if (NF) {
                    if (NF > ((FCHUNKS - 1) << FCH_LOG) - (ftail - fhead)) { failed = true; why = 4; break; }
                    if (!chunk_get(fchunk, fchunks_have, min((ftail + NF - 1) >> FCH_LOG, FCHUNKS - 1), sizeof(Far) << FCH_LOG)) { failed = true; why = 5; break; }
                    uint32_t fbad = 0;
                    for (uint32_t k = lane; k < NF; k += 64) {
                        const uint2 e = bk[Nb + k];
                        const uint32_t en = farEnd[k], prev = k ? farEnd[k - 1] : last_far_end;
                        if (en < prev) fbad = 1;  // the FIFO lives on non-decreasing ends
                        *far_rec(ftail + k) = Far{en, INF, e.y & 0xFFFFu, rb + (e.y >> 16)};
                    }
                    if (__ballot(fbad != 0) != 0) { failed = true; why = 6; break; }
                    last_far_end = bcast32(farEnd[NF - 1]);
                    if (next_head_end == 0xFFFFFFFFu && fhead + fin == ftail) next_head_end = bcast32(farEnd[0]);
                    ftail += NF;
                }
                fhead += fin;
                fhead_end = next_head_end;
                const uint32_t c0 = boff[nw], nc = Nb - c0;
                const uint32_t gnext = gw + N, rbn = gnext >= 0x8000u ? gnext - 0x8000u : 0u;  // the next window's base
                uint32_t cbad = 0;
                for (uint32_t sl = c0 + lane; sl < Nb; sl += 64) {  // lane = carried slot, re-based
                    const uint32_t y = bk[sl].y, g = rb + (y >> 16);
                    if (g < rbn) cbad = 1;
                    bank_y[kwin & 1u][sl - c0] = (y & 0xFFFFu) | ((g - rbn) << 16);
                    bank_rel[kwin & 1u][sl - c0] = (uint8_t)(brel[sl] - nw);
                }
                if (__ballot(cbad != 0) != 0) { failed = true; why = 7; break; }
                if (lane == 0) {
                    team->w0 = w0 + nw; team->gw = gw + N; team->ncarry = nc; team->fhead = fhead; team->ftail = ftail; team->last_far_end = last_far_end; team->fhead_end = fhead_end;
                    team->wbyte0 = wbyte_next; team->wlim = min(WIN, max(4u, fit_len)); team->nchunks_have = nchunks_have; team->fchunks_have = fchunks_have;
                }
                team_release();
                lds_put(&team->s_done, kwin + 1u);
                holding = false;
            }
            KW_T(5);
            KW_ARGS();
            // -- stage B: the tile list (lane = start position), then gather + sweep (kgpu_device.h: tiles_run), position by position in order
            const uint32_t lds0 = (uint32_t)(uintptr_t)(KGPU_LDS(uint8_t) *)lds;
            const uint32_t a_node = bcast32(lds0 + (uint32_t)((uint8_t *)node - lds)), a_bk = bcast32(lds0 + (uint32_t)((uint8_t *)bk - lds));
            uint64_t slowmask;   // positions with streamed predecessors: the any-shape step below, between the runs of tiles
            uint32_t eb_l;       // lane q: first tile of position q; lane nw: the end of the list
            {
                const bool inq = lane < nw;
                const uint32_t t0 = nb[inq ? lane : nw], T = inq ? nb[lane + 1] - t0 : 0u, p0 = boff[inq ? lane : nw], P = inq ? boff[lane + 1] - p0 : 0u;
                const uint32_t wn = inq ? wideN[lane] : 0u;
                eb_l = ebase[min(lane, nw)];
                slowmask = __ballot(inq && wn != 0 && T != 0);
                if (inq && wn == 0 && T) {
                    if (P == 0) {   // nothing ends here (lattice.rs:121-140 with an empty edges[pos]): its targets stay at INF (set by emit) with no predecessor -- no step
                        for (uint32_t k = 0; k < T; ++k) node[t0 + k].y = NONE16;
                    } else {
                        uint32_t k = eb_l;
                        const uint32_t kb = (P + 7u) >> 3;
                        for (uint32_t ta = 0; ta < T; ta += 8)
                            for (uint32_t b = 0; b < kb; ++b, ++k)
                                tiles[k] = make_uint2(tile_desc0(a_node + 8 * (t0 + ta), min(8u, T - ta), min(8u, P - 8 * b), b == 0, b == kb - 1), a_bk + 8 * (p0 + 8 * b));
                    }
                }
                if (lane == 0) node[N] = make_uint2(0u, 0u);   // what the lanes past a run of tiles gather for: row 0 of the matrix, never swept
            }
            wave_sync();
            KW_T(6);
            if constexpr (TEAM > 1) {
                if (!have_v) {   // the value token: every window before this one has been relaxed; the seeds' slots get their dp
                    bool stop = false;
                    for (;;) {
                        if (lds_get(&team->failed)) { stop = true; break; }
                        if (lds_get(&team->v_done) == kwin) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (stop) break;
                    team_acquire();
                    for (uint32_t sl = lane; sl < Nb; sl += 64) {
                        const uint32_t x = bk[sl].x;
                        if ((x & SEED_MARK) == SEED_MARK) {
                            // (every slot below Nb holds INF or a marker here -- emit wrote INF for this window's own nodes, the seeds step the markers -- so the
                            // pattern cannot be a real dp; the index is checked all the same: a stale slot must not become an out-of-bounds read of the FIFO)
                            const uint32_t idx = x & (SEED_FAR - 1u);
                            if ((x & SEED_FAR) ? idx < fin : idx < ncarry) bk[sl].x = (x & SEED_FAR) ? (uint32_t)far_rec(fhead_w + idx)->dp : bank_dp[pb][idx];
                        }
                    }
                    wave_sync();
                    have_v = true;
                }
            }
            {
                const uint8_t *connb = (const uint8_t *)d.conn;
                const uint2 null_tile = make_uint2((a_node + 8 * N) | TILE_FIRST, a_bk);
                for (uint32_t q = 0; q < nw;) {
                    const uint64_t rest = slowmask >> q;
                    const uint32_t qs = rest ? q + (uint32_t)__ffsll((unsigned long long)rest) - 1u : nw;   // the next any-shape position (nw: none)
                    const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)eb_l, (int)q), tb = (uint32_t)__builtin_amdgcn_readlane((int)eb_l, (int)qs);
                    if (tb > ta) tiles_run(tiles, ta, tb, null_tile, lane, a_bk, connb, true);
                    if (qs < nw) {
                        // any shape with streamed (wide) predecessors: SLOWT targets at a time, the lanes split the predecessors -- the bucket's in LDS, then the
                        // FIFO's; costs straight from the matrix (connection.rs:12-14); key = (total, node index): strict '<' over ascending insertion order (lattice.rs:125,136)
                        const uint32_t t0 = bcast32(nb[qs]), T = bcast32(nb[qs + 1]) - t0, p0 = bcast32(boff[qs]), P = bcast32(boff[qs + 1]) - p0;
                        const uint32_t wn = bcast32(wideN[qs]), wlo = bcast32(fcnt[qs]);
                        for (uint32_t tg = 0; tg < T; tg += SLOWT) {  // SLOWT targets share every load of a predecessor
                            const uint32_t nt8 = min(SLOWT, T - tg);
                            uint64_t key[SLOWT];
                            const uint8_t *col[SLOWT];
#pragma unroll
                            for (int k = 0; k < (int)SLOWT; ++k) {
                                key[k] = ~0ull;
                                col[k] = connb + node[t0 + tg + min((uint32_t)k, nt8 - 1)].y;   // the target's row of the matrix
                            }
                            for (uint32_t jj = lane; jj < P; jj += 64) {
                                const uint2 e = bk[p0 + jj];
                                const uint32_t gi = rb + (e.y >> 16), r2 = e.y & 0xFFFFu;
#pragma unroll
                                for (int k = 0; k < (int)SLOWT; ++k)
                                    if ((uint32_t)k < nt8) {
                                        const int32_t v = (int32_t)e.x + (int32_t)*(const int16_t *)(col[k] + r2);
                                        const uint64_t ck = ((uint64_t)((uint32_t)v ^ 0x80000000u) << 32) | gi;
                                        key[k] = ck < key[k] ? ck : key[k];
                                    }
                            }
                            for (uint32_t f = wlo + lane; f < wlo + wn; f += 64) {
                                const Far e = *far_rec(f);
                                const uint32_t r2 = e.right & 0xFFFFu;
#pragma unroll
                                for (int k = 0; k < (int)SLOWT; ++k)
                                    if ((uint32_t)k < nt8) {
                                        const int32_t v = e.dp + (int32_t)*(const int16_t *)(col[k] + r2);
                                        const uint64_t ck = ((uint64_t)((uint32_t)v ^ 0x80000000u) << 32) | e.node;
                                        key[k] = ck < key[k] ? ck : key[k];
                                    }
                            }
#pragma unroll
                            for (int k = 0; k < (int)SLOWT; ++k)
                                if ((uint32_t)k < nt8) {
                                    const uint64_t kk = wave_min_u64(key[k]);
                                    if (lane == 0) {
                                        const uint32_t cs = node[t0 + tg + k].x;
                                        int32_t dpv = INF; uint32_t prv = NONE16;
                                        if (kk != ~0ull) {
                                            const int32_t tot = (int32_t)((uint32_t)(kk >> 32) ^ 0x80000000u) + (int32_t)(int16_t)cs;
                                            if (tot < INF) { dpv = tot; prv = (uint32_t)kk - rb; }
                                        }
                                        node[t0 + tg + k].y = prv & 0xFFFFu;
                                        bk[cs >> SLOT_SHIFT].x = (uint32_t)dpv;
                                    }
                                }
                            wave_sync();
                        }
                    }
                    q = qs + 1;
                }
            }
            wave_sync();
            KW_T(7);

            KW_ARGS();
            if constexpr (TEAM > 1) {
                if (!have_v) break;   // (another wavefront failed the sentence while this one waited for its value token)
                // ---- this window is relaxed: the dp of what it carries on and of its far entries, then the value token
                const uint32_t c0 = boff[nw];
                for (uint32_t sl = c0 + lane; sl < Nb; sl += 64) bank_dp[kwin & 1u][sl - c0] = bk[sl].x;
                for (uint32_t k = lane; k < NF; k += 64) far_rec(ftail_w + k)->dp = (int32_t)bk[Nb + k].x;
                team_release();
                lds_put(&team->v_done, kwin + 1u);
            }
            // -- flush: node records to HBM, far-out entries to the FIFO, the buckets beyond the window to the carry list
            if constexpr (TEAM == 1) if (!chunk_get(nchunk, nchunks_have, (gw + N - 1) >> NCH_LOG, (4u + (uint32_t)sizeof(NodeRec)) << NCH_LOG)) { failed = true; why = 3; break; }
            for (uint32_t t = lane; t < N; t += 64) {
                const uint32_t p = node[t].y & 0xFFFFu, st = nStart[t];
                const uint32_t gp = p == NONE16 ? NONE : rb + p;
                NodeRec rec;
                rec.sid = nSid[t];
                rec.start = w0 + st;
                *node_pre(gw + t) = gp;
                *node_rec(gw + t) = rec;
                if (rec.sid == 0 && w0 + st == C) eos_pre = gp;  // (only EOS has sid 0: BOS is never a target)
            }
            if (nw > nwc) eos_pre = bcast32((uint32_t)__shfl((int)eos_pre, (int)((N - 1) & 63u), 64));  // the lane that wrote node N - 1
            if constexpr (TEAM > 1) {
                if (nw > nwc) lds_put(&team->eos_pre, eos_pre);
                wT += wTw; wE += wEw;
                wave_sync();
                KW_T(8);
                kwin += (uint32_t)TEAM;
                continue;
            }
            if (NF) {
                if (NF > ((FCHUNKS - 1) << FCH_LOG) - (ftail - fhead)) { failed = true; why = 4; break; }
                if (!chunk_get(fchunk, fchunks_have, min((ftail + NF - 1) >> FCH_LOG, FCHUNKS - 1), sizeof(Far) << FCH_LOG)) { failed = true; why = 5; break; }
                uint32_t fbad = 0;
                for (uint32_t k = lane; k < NF; k += 64) {
                    const uint2 e = bk[Nb + k];
                    const uint32_t en = farEnd[k], prev = k ? farEnd[k - 1] : last_far_end;
                    if (en < prev) fbad = 1;  // the FIFO lives on non-decreasing ends
                    *far_rec(ftail + k) = Far{en, (int32_t)e.x, e.y & 0xFFFFu, rb + (e.y >> 16)};
                }
                if (__ballot(fbad != 0) != 0) { failed = true; why = 6; break; }
                last_far_end = bcast32(farEnd[NF - 1]);
                if (next_head_end == 0xFFFFFFFFu && fhead + fin == ftail) next_head_end = bcast32(farEnd[0]);   // the FIFO was (or has just become) empty: these head it
                ftail += NF;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the next window may already take some of these from the FIFO
            }
            fhead += fin;
            fhead_end = next_head_end;
            {
                const uint32_t c0 = boff[nw], nc = Nb - c0;
                const uint32_t gnext = gw + N, rbn = gnext >= 0x8000u ? gnext - 0x8000u : 0u;  // the next window's base
                uint2 *c8 = carry8(nc);
                uint8_t *cr = crel(nc);
                uint32_t cbad = 0;
                for (uint32_t sl = c0 + lane; sl < Nb; sl += 64) {  // lane = carried slot, re-based
                    const uint2 e = bk[sl];
                    const uint32_t g = rb + (e.y >> 16);
                    if (g < rbn) cbad = 1;
                    c8[sl - c0] = make_uint2(e.x, (e.y & 0xFFFFu) | ((g - rbn) << 16));
                    cr[sl - c0] = (uint8_t)(brel[sl] - nw);
                }
                if (__ballot(cbad != 0) != 0) { failed = true; why = 7; break; }
                ncarry = nc;
            }
            wlim = min(WIN, max(4u, fit_len));   // the next window: as long as fills 7/8 of the LDS at this window's density
            wT += wTw; wE += wEw;
            wbyte0 = wbyte_next;
            staged = false;
            wave_sync();  // (no workgroup barrier: the FIFO entries just written are read back by this same wavefront, in program order)
            KW_T(8);
            gw += N;
            w0 += nw;
        }
        uint32_t gw_end = gw;
        if constexpr (TEAM > 1) {
            if (failed && lane == 0) { team->why = why; __hip_atomic_store(&team->failed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if constexpr (PROF) {
                wT = wave_sum(wT); wE = wave_sum(wE);
                if (lane == 0) { atomicAdd(&team->wT, wT); atomicAdd(&team->wE, wE); }
            }
            team_release();
            __syncthreads();   // every wavefront is through with its windows: node records, far entries and the team's state are complete
            team_acquire();
            if (wave != 0) continue;   // the first wavefront alone chases the path and writes the tokens (the others wait at the next sentence's first barrier)
            failed = bcast32(team->failed) != 0; why = bcast32(team->why);
            gw_end = bcast32(team->gw); eos_pre = bcast32(team->eos_pre);
            wT = bcast32(team->wT); wE = bcast32(team->wE);
        }
        if (failed) {
            if ((why == 3 || why == 5) && __hip_atomic_load(&a.ctl->arena_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (lane == 0) { a.status[s] = KGPU_SENT_NO_SCRATCH; a.tok_count[s] = 0; }  // the scratch arena was too small: the host grows it and reruns the batch
            } else fail(s);
            if (lane == 0 && !PROF) atomicAdd(&a.ctl->phase[why < 10 ? why : 9], 1ull);
            setup_fence();
            continue;
        }

        KW_ARGS();
        // ---- backtrace (lattice.rs:144-153): chase `pre` through windows of node records staged in LDS ----
        KW_T(8);
        const uint32_t Ntot = gw_end;  // BOS + every node; EOS is node Ntot - 1
        uint32_t K = 0;
        {
            // `pre` of a range of nodes staged in LDS, lane 0 chases the chain through it; the path it finds is parked in LDS too (PATHL entries at the
            // top) and written out PATHL at a time by all lanes -- a global store inside the one-lane loop made every step wait for its issue
            uint32_t *win = (uint32_t *)(lds + off0);
            const uint32_t Wall = (lds_bytes - off0) / 4, PATHL = min(512u, Wall / 4), Wn = Wall - PATHL;
            uint32_t *lpath = win + Wn;
            uint32_t pos = Ntot - 1, nl = 0;
            bool done = eos_pre == NONE;  // EOS unreachable: empty Vec (lattice.rs:144-153)
            if (!done) { if (lane == 0) lpath[0] = pos; nl = 1; pos = eos_pre; }
            while (!done) {
                // pos: the node whose record is needed next (a predecessor always has a smaller index)
                const uint32_t wlo = pos >= Wn - 1 ? pos - (Wn - 1) : 0;
                for (uint32_t i = wlo + lane; i <= pos; i += 64) win[i - wlo] = i ? *node_pre(i) : NONE;
                wave_sync();
                for (;;) {
                    uint32_t npos = pos, nnl = nl, fin2 = 0;
                    if (lane == 0) {
                        // (one lane's walk at the whole wavefront's four cycles an instruction: spelled out -- the compiler's version of this loop is 32 instructions
                        // a step, this one 13: the address, the read, the end test, the path's store, the window test, the count.  `cap`: room in the parked path,
                        // and the bound C + 1 on a path's length, which a chain of falling indices cannot reach before its end: then the walk is over as well.)
                        const uint32_t room = PATHL - nnl, left = K + nnl > C ? 0u : C + 1 - K - nnl, cap = min(room, left);
                        const uint32_t a_lds = (uint32_t)(uintptr_t)(KGPU_LDS(uint8_t) *)lds;
                        const uint32_t a_rel = a_lds + (uint32_t)((uint8_t *)win - lds) - 4u * wlo;   // LDS address of win[i - wlo] = a_rel + 4 i
                        uint32_t lp = a_lds + (uint32_t)((uint8_t *)(lpath + nnl) - lds), n = 0, ad, pr;
                        if (cap == 0) fin2 = 1;
                        else {
                            asm volatile(
                                "s_mov_b32 %[n], 0\n\t"
                                "s_mov_b32 %[fin], 0\n"
                                "Lwb_loop%=:\n\t"
                                "v_lshl_add_u32 %[ad], %[pos], 2, %[rel]\n\t"
                                "ds_read_b32 %[pr], %[ad]\n\t"
                                "s_waitcnt lgkmcnt(0)\n\t"
                                "v_cmp_eq_u32_e32 vcc, -1, %[pr]\n\t"
                                "s_cbranch_vccnz Lwb_fin%=\n\t"
                                "ds_write_b32 %[lp], %[pos]\n\t"
                                "v_add_u32_e32 %[lp], 4, %[lp]\n\t"
                                "v_mov_b32_e32 %[pos], %[pr]\n\t"
                                "s_add_u32 %[n], %[n], 1\n\t"
                                "v_cmp_gt_u32_e32 vcc, %[wlo], %[pr]\n\t"
                                "s_cbranch_vccnz Lwb_out%=\n\t"
                                "s_cmp_lt_u32 %[n], %[cap]\n\t"
                                "s_cbranch_scc1 Lwb_loop%=\n\t"
                                "s_branch Lwb_out%=\n"
                                "Lwb_fin%=:\n\t"
                                "s_mov_b32 %[fin], 1\n"
                                "Lwb_out%=:\n\t"
                                "s_waitcnt lgkmcnt(0)"
                                : [n] "=&s"(n), [fin] "=&s"(fin2), [pos] "+v"(npos), [lp] "+v"(lp), [ad] "=&v"(ad), [pr] "=&v"(pr)
                                : [rel] "s"(bcast32(a_rel)), [wlo] "s"(bcast32(wlo)), [cap] "s"(bcast32(cap))
                                : "vcc", "scc", "memory");
                            nnl += n;
                            if (!fin2 && n == cap && left <= room) fin2 = 1;   // the bound, not the parking space: over
                        }
                    }
                    pos = bcast32(npos); nl = bcast32(nnl); done = bcast32(fin2) != 0;
                    wave_sync();
                    if (nl == PATHL || done) {
                        for (uint32_t k = lane; k < nl; k += 64) path[K + k] = lpath[k];
                        K += nl; nl = 0;
                        wave_sync();
                    }
                    if (done || pos < wlo) break;
                }
            }
        }
        K = bcast32(K);
        const uint64_t ts = b0 - a.offsets[0] + s;
        setup_fence();
        for (uint32_t k = lane; k < K; k += 64) {  // Node -> Token (tokenizer.rs:22-43); a word ends where its successor starts
            const NodeRec r = *node_rec(path[K - 1 - k]);
            kgpu_token tk;
            if (r.sid == 0) { tk.id = 0; tk.cls = KGPU_CLASS_DUMMY; tk.position = B; tk.start = C; tk.end = C + 3; tk.byte_len = 0; }
            else {   // a word ends where its successor starts; the byte offsets come from the per-character records of the decode pass ([C] = {B, 0})
                const NodeRec nx = *node_rec(path[K - 2 - k]);
                const uint32_t b0s = crec[r.start].x & 0xFFFFFFu, b1s = crec[nx.start].x & 0xFFFFFFu;
                tk.id = r.sid > 0 ? r.sid : -r.sid;
                tk.cls = r.sid > 0 ? KGPU_CLASS_KNOWN : KGPU_CLASS_UNKNOWN;
                tk.position = b0s; tk.start = r.start; tk.end = nx.start; tk.byte_len = b1s - b0s;
            }
            a.stage[ts + k] = tk;
        }
        if (lane == 0) { a.status[s] = KGPU_SENT_OK; a.tok_count[s] = K; }
        KW_T(9);
        if constexpr (PROF) {
            if constexpr (TEAM == 1) { wT = wave_sum(wT); wE = wave_sum(wE); }
            accW[0] += 1; accW[1] += B; accW[2] += C; accW[3] += wT; accW[4] += Ntot - 1; accW[5] += wE; accW[6] += K;
        }
        setup_fence();
    }
    if (PROF && lane == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&a.ctl->work[k], (unsigned long long)accW[k]);
        for (int k = 0; k < 10; ++k) atomicAdd(&a.ctl->phase[k], (unsigned long long)tph[k]);
    }
}

// LDS of one workgroup of TEAM wavefronts (the kernel's own carve: chunk tables, team state, two carry banks, TEAM regions)
static uint32_t team_lds_bytes(uint32_t lds_bytes, int team) {
    const uint32_t shared0 = 4 * NCHUNKS + 4 * FCHUNKS;
    return team > 1 ? ((shared0 + (uint32_t)sizeof(TeamState) + 15u) & ~15u) + 2u * 9u * CCAP + (uint32_t)team * lds_bytes : lds_bytes;
}

int window_workgroups_per_cu(uint32_t lds_bytes) {
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void *)k_tokenize_window<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_tokenize_window<false, 1>, 64, (size_t)lds_bytes) != hipSuccess) return 0;
    return n;
}
// ... of the form with `team` wavefronts per sentence (0: it does not fit)
template <int TEAM>
static int team_per_cu(uint32_t lds_bytes) {
    const uint32_t tb = team_lds_bytes(lds_bytes, TEAM);
    if (tb > 64 * 1024 && hipFuncSetAttribute((const void *)k_tokenize_window<false, TEAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_tokenize_window<false, TEAM>, 64 * TEAM, (size_t)tb) != hipSuccess) return 0;
    return n;
}
int window_team_workgroups_per_cu(uint32_t lds_bytes, int team) { return team == 2 ? team_per_cu<2>(lds_bytes) : team == 3 ? team_per_cu<3>(lds_bytes) : team == 4 ? team_per_cu<4>(lds_bytes) : 0; }

template <bool PROF, int TEAM>
static int launch_window_inst(const WinArgs &wa, int n_workgroups, void *stream) {
    const uint32_t tb = team_lds_bytes(wa.lds_bytes, TEAM);
    if (tb > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_tokenize_window<PROF, TEAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((k_tokenize_window<PROF, TEAM>), dim3((unsigned)n_workgroups), dim3(64 * TEAM), tb, (hipStream_t)stream, wa);
    return (int)hipGetLastError();
}

// team: wavefronts per sentence (1; 2-4: the host picks that when the work list is short against the chip's slots)
int launch_tokenize_window(const DictView &d, const BatchArgs &a, const WorkIO &io, uint32_t lds_bytes, int n_workgroups, int team, void *stream, bool claim) {
    const WinArgs wa{d, a, io, lds_bytes, (team <= 1 && claim) ? 1u : 0u};
    if (team == 2) return a.count_work ? launch_window_inst<true, 2>(wa, n_workgroups, stream) : launch_window_inst<false, 2>(wa, n_workgroups, stream);
    if (team == 3) return a.count_work ? launch_window_inst<true, 3>(wa, n_workgroups, stream) : launch_window_inst<false, 3>(wa, n_workgroups, stream);
    if (team == 4) return a.count_work ? launch_window_inst<true, 4>(wa, n_workgroups, stream) : launch_window_inst<false, 4>(wa, n_workgroups, stream);
    return a.count_work ? launch_window_inst<true, 1>(wa, n_workgroups, stream) : launch_window_inst<false, 1>(wa, n_workgroups, stream);
}

}  // namespace kgpu

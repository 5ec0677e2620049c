// kanpyo_amd/csrc/kgpu_kernels.hip -- hand-written gfx950 (CDNA4) kernels for
// the Tokenizer::tokenize hot path (reference src/tokenizer.rs:16-45).
//
// One 64-lane wavefront owns one sentence from UTF-8 bytes to token records:
//   phase 0  UTF-8 decode + validation, char -> byte offset, char category
//            (char_category_def.rs:33-38)
//   phase 1  lattice COUNT: one lane per start position walks the double array
//            (trie/da.rs:155-182), expands duplicates (index.rs:46-51), decides the
//            unknown-word span (lattice.rs:42-99); per-position node counts and
//            per-end-position bucket counts
//   phase 2  wave prefix sums -> node numbering in reference insertion order
//            (lattice.rs:105-110) and bucket offsets (= Lattice.edges, lattice.rs:9)
//   phase 3  lattice EMIT: same walk, nodes written to their numbered slots
//   phase 4  Viterbi sweep over start positions (lattice.rs:116-142): all nodes
//            starting at q share the predecessor bucket edges[q]; lanes take
//            targets, the bucket is broadcast; min is lexicographic on
//            (total, predecessor index) = strict '<' over ascending insertion order
//   phase 5  backtrace (lattice.rs:144-153) + Node -> Token (tokenizer.rs:22-43)
// Integer/indexing work only: no MFMA anywhere.
//
// This file holds the GENERAL kernel: every per-sentence array lives in a bump-allocated HBM scratch slab and every
// step of the Viterbi chain is a global-memory round trip -- no LDS at all, so its workgroups start on a CU whose LDS is
// fully taken, and any sentence length / lattice shape works.  It is the LAST RESORT of the launch chain (what the LDS-resident
// pool kernel, kgpu_pool.hip, and the windowed kernel, kgpu_window.hip, cannot hold) and the kernel behind kgpu_lattice_dump: its
// slabs ARE the lattice.  (Round 2-3 also instantiated it with an LDS-blocked sweep as "the long-sentence kernel"; the windowed kernel
// took that place in round 4.)  Also here: scan / compaction kernels and the launch chain.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kgpu_device.h"

namespace kgpu {

using namespace dev;

namespace {

constexpr uint32_t GMAXM = 8;  // trie matches parked per start position by the count walk

}  // namespace

__global__ __launch_bounds__(64) void k_tokenize_general(DictView d, BatchArgs a, WorkIO io, uint32_t stop_after /* ablation timing only */) {
    const uint32_t lane = threadIdx.x;
    Slab sa{nullptr, 0}, sn{nullptr, 0};
    const int32_t base_root = d.da[1].base;
    uint64_t accW[7] = {0, 0, 0, 0, 0, 0, 0};  // work counters of this workgroup, flushed once at exit

    for (uint32_t iter = 0;; ++iter) {
        uint64_t s = 0;
        if (!work_next(io, a, iter, s)) break;

        const uint64_t b0 = a.offsets[s];
        const uint32_t B = (uint32_t)(a.offsets[s + 1] - b0);
        const uint8_t *text = a.utf8 + b0;

        // ---- slab A: per-char arrays (C <= B) --------------------------------
        const uint64_t na = (uint64_t)B + 4;
        if (!slab_ensure(sa, na * (28 + 5 * GMAXM) + 64, a, lane)) {
            if (lane == 0) { a.status[s] = KGPU_SENT_NO_SCRATCH; a.tok_count[s] = 0; }
            continue;
        }
        uint32_t *cbyte = (uint32_t *)sa.ptr;  // char -> byte offset, [C] = B
        uint32_t *uspan = cbyte + na;          // unknown span in chars (0 = none)
        uint32_t *nb = uspan + na;             // node count, then first node index, per start
        uint32_t *boff = nb + na;              // bucket count, then offset, per end
        uint32_t *bfill = boff + na;           // bucket fill cursor
        uint32_t *path = bfill + na;           // backtrace
        uint32_t *mid = path + na;             // [na][GMAXM] trie ids parked by the count walk (the emit phase re-walks nothing)
        uint16_t *cp16 = (uint16_t *)(mid + na * GMAXM);  // BMP code point (0xFFFF: not BMP)
        uint8_t *ccat = (uint8_t *)(cp16 + na);
        uint8_t *mcnt = ccat + na;             // parked matches per start position (0xFF: more than GMAXM -> re-walk)
        uint8_t *mnch = mcnt + na;             // [na][GMAXM] match length in chars

        // ---- phase 0: decode ------------------------------------------------
        const bool ct = d.da2 != nullptr;  // character-level trie (kgpu_chartrie.cpp): the walks read character codes, not bytes; cp16[] then holds the codes
        uint32_t C = 0, bad = 0, lensum = 0;
        for (uint32_t k0 = 0; k0 < B; k0 += 64) {
            uint32_t k = k0 + lane;
            uint32_t b = k < B ? text[k] : 0x80u;
            bool start = k < B && (b & 0xC0) != 0x80;
            uint64_t m = __ballot(start);
            uint32_t ci = C + __popcll(m & ((1ull << lane) - 1));
            if (start) {
                uint32_t cp;
                const uint32_t l = utf8_decode_lead(b, k, B, [&](uint32_t kk) -> uint32_t { return text[kk]; }, cp, bad);
                lensum += l;
                cbyte[ci] = k;
                uint32_t code = cp < 0xFFFFu ? cp : 0xFFFFu;
                if (ct) code = cp < 0xFFFFu ? (uint32_t)d.crec[cp].code : (d.n_nb ? ct_code_nonbmp(d, cp) : 0xFFFFu);
                cp16[ci] = (uint16_t)code;
                // char_category_def.rs:33-38: table[ch] if in range else table[0]
                ccat[ci] = bad ? 0 : (cp < d.cat_len ? d.cat[cp] : d.cat[0]);
            }
            C += __popcll(m);
        }
        lensum = bcast32(wave_sum(lensum));  // keep every early exit wave-uniform (SGPR) for the compiler
        if (__ballot(bad != 0) != 0 || lensum != B) {  // stray continuation bytes leave lensum < B
            if (lane == 0) { a.status[s] = KGPU_SENT_INVALID_UTF8; a.tok_count[s] = 0; }
            continue;
        }
        if (lane == 0) { cbyte[C] = B; if (ct) cp16[C] = 0xFFFFu; }
        uint32_t *cnt_e = boff, *fill_e = bfill;  // per-end-position counters / fill cursors (global atomics)
        const uint8_t *wtext = text;
        auto code_at = [&](uint32_t j) -> uint32_t { return (uint32_t)cp16[min(j, C)]; };  // ([C]: none)
        for (uint32_t e = lane; e < C + 3; e += 64) { cnt_e[e] = 0; bfill[e] = 0; }
        __syncthreads();

        // ---- phase 1: count ----------------------------------------------------
        const int nchunks = (int)((C + 63) / 64);
        uint32_t carry_end = C;
        uint32_t wT = 0, wE = 0;  // work counters (only reduced when a.count_work)
        // two chunks of 64 start positions per round, walked side by side (da_walk_first2): a chunk costs its longest
        // walk, a pair the longer of the two -- not their sum (the count walk is 21-28 % of this kernel)
        for (int ch = nchunks - 1; ch >= 0; ch -= 2) {
            uint32_t iX[2], catX[2], runX[2], cntX[2] = {0, 0}, mX[2] = {0, 0};
            bool actX[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {  // x = 0: chunk ch, x = 1: chunk ch - 1 (run ends carry from the later chunk to the earlier)
                const bool have = ch - x >= 0;
                const uint32_t i = have ? (uint32_t)(ch - x) * 64 + lane : 0xFFFFFFF0u;
                const bool active = have && i < C;
                const uint32_t cat = active ? ccat[i] : 0x1FFu;
                const uint32_t ncat = (have && i + 1 < C) ? ccat[i + 1] : 0x2FFu;
                const uint64_t bm = __ballot(active && ncat != cat);
                const uint64_t rest = bm >> lane;
                const uint32_t run_end = rest ? i + (uint32_t)__ffsll((unsigned long long)rest) : carry_end;
                if (have) carry_end = bcast32(run_end);
                iX[x] = i; actX[x] = active; catX[x] = cat; runX[x] = run_end;
            }
            auto on_match = [&](int x) {
                return [&, x](uint32_t id, uint32_t nch, uint32_t dup) {
                    const uint32_t i = iX[x];
                    uint32_t &m = mX[x];
                    if (m < GMAXM && nch < 256) { mid[(size_t)i * GMAXM + m] = id; mnch[(size_t)i * GMAXM + m] = (uint8_t)nch; }
                    else m = 0x100;  // does not fit the parking area: the emit phase walks again
                    ++m;
                    uint32_t nrec = 1u + (dup != NONE ? dup : (uint32_t)d.morph[id - 1].dup);  // index.rs:46-51
                    cntX[x] += nrec;
                    atomicAdd(&cnt_e[i + nch], nrec);
                };
            };
            uint32_t cpX[2], kbX[2], knX[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) { cpX[x] = actX[x] ? cp16[iX[x]] : 0xFFFFu; kbX[x] = actX[x] ? cbyte[iX[x]] : 0u; knX[x] = actX[x] ? cbyte[iX[x] + 1] : 0u; }
            if (ct) {
                ct_walk2(d, actX[0], [&](uint32_t dep) { return code_at(iX[0] + dep); }, on_match(0), actX[1], [&](uint32_t dep) { return code_at(iX[1] + dep); }, on_match(1));
                if (a.count_work == 1u) {  // the reference's byte steps (work counters; not with KGPU_PROFILE_NO_T)
#pragma unroll
                    for (int x = 0; x < 2; ++x) if (actX[x]) wT += da_walk(d, text, kbX[x], B, base_root, [](uint32_t, uint32_t, uint32_t) {});
                }
            } else
            wT += da_walk_first2(d, wtext, B, actX[0] && cpX[0] != 0xFFFFu, cpX[0], kbX[0], knX[0], on_match(0),
                                 actX[1] && cpX[1] != 0xFFFFu, cpX[1], kbX[1], knX[1], on_match(1));
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                if (!actX[x]) continue;
                const uint32_t i = iX[x];
                if (!ct && cpX[x] == 0xFFFFu) wT += da_walk(d, wtext, kbX[x], B, base_root, on_match(x));  // first character outside the BMP: no table entry
                uint32_t cnt = cntX[x];
                const uint32_t m = mX[x];
                mcnt[i] = (uint8_t)(m > GMAXM ? 0xFFu : m);
                const CatInfo ci = d.cinfo[catX[x]];
                uint32_t span = 0;
                // lattice.rs:54: !matched_known || invoke_list[cat]; lattice.rs:87-92: no unk entry -> nothing
                if ((cnt == 0 || (ci.flags & CAT_INVOKE)) && (ci.flags & CAT_HAS_UNK) && ci.unk_count) {
                    span = 1;
                    if (ci.flags & CAT_GROUP) {  // lattice.rs:66-84
                        uint32_t r = runX[x] - i;
                        span = r < MAX_UNKNOWN_LEN ? r : MAX_UNKNOWN_LEN;
                    }
                    cnt += ci.unk_count;
                    atomicAdd(&cnt_e[i + span], ci.unk_count);
                }
                uspan[i] = span;
                nb[i] = cnt;
            }
        }
        if (lane == 0) {
            nb[C] = 1;      // EOS starts at C (lattice.rs:165-175)
            nb[C + 1] = 0;
            atomicAdd(&cnt_e[0], 1u);  // BOS ends at 0 (lattice.rs:156-164)
        }
        __syncthreads();

#define KGPU_GSTOP(k) if (stop_after == (k)) { if (lane == 0) { a.status[s] = KGPU_SENT_TRUNCATED; a.tok_count[s] = 0; } continue; }
        KGPU_GSTOP(3)
        // ---- phase 2: prefix sums ------------------------------------------------
        uint32_t ncarry = 1, bcarry = 0;  // node 0 is BOS
        for (uint32_t i0 = 0; i0 < C + 2; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint32_t v = i < C + 2 ? nb[i] : 0;
            const uint32_t w = i < C + 2 ? ld_l2(&boff[i]) : 0;  // updated by L2 atomics
            const uint32_t vs = wave_incl_scan(v, lane), ws = wave_incl_scan(w, lane);
            if (i < C + 2) { nb[i] = ncarry + vs - v; boff[i] = bcarry + ws - w; }
            ncarry += __shfl(vs, 63, 64);
            bcarry += __shfl(ws, 63, 64);
        }
        const uint32_t N = bcast32(ncarry);  // BOS + words + EOS (scalarised: the slab test must be wave-uniform)
        __syncthreads();

        // ---- slab N: per-node arrays -----------------------------------------------
        if (!slab_ensure(sn, (uint64_t)N * 44 + 64, a, lane)) {
            if (lane == 0) { a.status[s] = KGPU_SENT_NO_SCRATCH; a.tok_count[s] = 0; }
            continue;
        }
        uint4 *nodeA = (uint4 *)sn.ptr;   // {left | right << 16, cost, bucket slot, signed id}
        uint4 *bucket = nodeA + N;        // {dp, right_id, node index, -}
        uint2 *nodeB = (uint2 *)(bucket + N);  // {start char, end char}
        uint32_t *pre = (uint32_t *)(nodeB + N);

        // ---- phase 3: emit -----------------------------------------------------------
        for (uint32_t i = lane; i < C; i += 64) {
            uint32_t t = nb[i];
            auto emit_match = [&](uint32_t id, uint32_t nch, uint32_t /*dup*/ = 0) {
                const uint32_t end = i + nch;
                const Morph8 m0 = d.morph[id - 1];  // first record: carries the duplicate count (index.rs:46-51)
                const uint32_t nrec = 1u + m0.dup;
                for (uint32_t r = 0; r < nrec; ++r) {  // lattice.rs:177-188
                    const Morph8 m = r ? d.morph[id - 1 + r] : m0;
                    const uint32_t slot = boff[end] + atomicAdd(&fill_e[end], 1u);
                    nodeA[t] = make_uint4((uint16_t)m.left | ((uint32_t)(uint16_t)m.right << 16),
                                          (uint32_t)(int32_t)m.cost, slot, id + r);
                    nodeB[t] = make_uint2(i, end);
                    bucket[slot] = make_uint4((uint32_t)INF, (uint16_t)m.right, t, 0);  // dp is filled in by the sweep
                    ++t;
                }
            };
            const uint32_t nm = mcnt[i];
            if (nm == 0xFFu) {
                if (ct) ct_walk(d, 1, d.da2[1].base, 0, [&](uint32_t dep) { return code_at(i + dep); }, emit_match, 0u);
                else da_walk(d, wtext, cbyte[i], B, base_root, emit_match);
            }
            else for (uint32_t m = 0; m < nm; ++m) emit_match(mid[(size_t)i * GMAXM + m], mnch[(size_t)i * GMAXM + m]);
            const uint32_t span = uspan[i];
            if (span) {  // lattice.rs:87-97,190-201
                const CatInfo ci = d.cinfo[ccat[i]];
                const uint32_t end = i + span;
                for (uint32_t r = 0; r < ci.unk_count; ++r) {
                    const Morph8 m = d.unk_morph[ci.unk_first - 1 + (int32_t)r];
                    const uint32_t slot = boff[end] + atomicAdd(&fill_e[end], 1u);
                    nodeA[t] = make_uint4((uint16_t)m.left | ((uint32_t)(uint16_t)m.right << 16),
                                          (uint32_t)(int32_t)m.cost, slot,
                                          (uint32_t)(-(ci.unk_first + (int32_t)r)));
                    nodeB[t] = make_uint2(i, end);
                    bucket[slot] = make_uint4((uint32_t)INF, (uint16_t)m.right, t, 0);
                    ++t;
                }
            }
        }
        if (lane == 0) {
            nodeA[N - 1] = make_uint4(d.eos_left, 0, NONE, 0);  // EOS: Morph(0,0,0), id 0 (context id 0 in its ranked numbering)
            nodeB[N - 1] = make_uint2(C, C);
            bucket[0] = make_uint4(0, d.bos_right, 0, 0);  // BOS: dp None -> 0 (lattice.rs:127)
            pre[0] = NONE;
        }
        __syncthreads();

        KGPU_GSTOP(5)
        // ---- phase 4: Viterbi sweep ---------------------------------------------------
        // one position through global memory (lattice.rs:116-142): lanes take targets, every lane
        // walks the whole predecessor bucket
        auto global_step = [&](uint32_t q) {
            const uint32_t t0 = nb[q], t1 = nb[q + 1], p0 = boff[q], P = boff[q + 1] - p0;
            wE += (lane == 0) ? P * (t1 - t0) : 0;
            if (P >= 48) {
                // wide bucket (e.g. every unknown word of a 1024-char run ends at the same position): the lanes
                // share one target at a time and split its predecessors -- coalesced bucket reads, P/64 steps
                for (uint32_t t = t0; t < t1; ++t) {
                    const uint4 na_ = nodeA[t];
                    const int16_t *col = d.conn + (size_t)d.conn_rows * (na_.x & 0xFFFFu);
                    int32_t best = 0x7FFFFFFF;
                    uint32_t bidx = NONE;
                    for (uint32_t j = lane; j < P; j += 64) {
                        const uint4 e = bucket[p0 + j];
                        const int32_t v = (int32_t)e.x + (int32_t)col[e.y];
                        if (v < best || (v == best && e.z < bidx)) { best = v; bidx = e.z; }
                    }
                    const int32_t vmin = wave_min_i32(best);
                    const uint32_t nmin = wave_min_u32(best == vmin ? bidx : NONE);
                    if (lane == 0) {
                        const int32_t tot = vmin + (int32_t)na_.y;  // min(.., INF) then strict '<' INF
                        const bool ok = tot < INF;
                        pre[t] = ok ? nmin : NONE;
                        if (na_.z != NONE) bucket[na_.z].x = (uint32_t)(ok ? tot : INF);
                        else if (a.dump_lattice) a.ctl->dump[6] = (unsigned long long)(uint32_t)(ok ? tot : INF);  // EOS has no bucket entry
                    }
                }
                __syncthreads();
                return;
            }
            for (uint32_t t = t0 + lane; t < t1; t += 64) {
                const uint4 na_ = nodeA[t];
                const int16_t *col = d.conn + (size_t)d.conn_rows * (na_.x & 0xFFFFu);
                int32_t best = 0x7FFFFFFF;
                uint32_t bidx = NONE;
                for (uint32_t j = 0; j < P; ++j) {
                    const uint4 e = bucket[p0 + j];
                    const int32_t v = (int32_t)e.x + (int32_t)col[e.y];
                    if (v < best || (v == best && e.z < bidx)) { best = v; bidx = e.z; }
                }
                int32_t dpv = INF;
                uint32_t prv = NONE;
                if (P) {
                    const int32_t tot = best + (int32_t)na_.y;  // min(.., INF) then strict '<' INF
                    if (tot < INF) { dpv = tot; prv = bidx; }
                }
                pre[t] = prv;
                if (na_.z != NONE) bucket[na_.z].x = (uint32_t)dpv;
                else if (a.dump_lattice) a.ctl->dump[6] = (unsigned long long)(uint32_t)dpv;  // EOS has no bucket entry
            }
            __syncthreads();
        };
        for (uint32_t q = 0; q <= C; ++q) global_step(q);
        wE = wave_sum(wE);

        KGPU_GSTOP(7)
        // ---- phase 5: backtrace + tokens -----------------------------------------------
        uint32_t K = 0;
        if (lane == 0) {
            uint32_t pos = N - 1, pr;
            while ((pr = pre[pos]) != NONE && K <= C) { path[K++] = pos; pos = pr; }  // K <= C + 1 always; bound the walk anyway
        }
        K = bcast32(K);
        // staging slot of the sentence: K <= C + 1 <= B + 1 tokens always fit at b0 + s
        // (no cursor atomics: a single hot word serialises ~90 sentences/us chip-wide)
        const uint64_t ts = b0 - a.offsets[0] + s;
        __syncthreads();
        {
            for (uint32_t k = lane; k < K; k += 64) {
                const uint32_t t = path[K - 1 - k];
                const int32_t sid = (int32_t)nodeA[t].w;
                const uint2 se = nodeB[t];
                kgpu_token tk;
                if (sid == 0) {  // Dummy -> "EOS" (tokenizer.rs:27-28,34)
                    tk.id = 0; tk.cls = KGPU_CLASS_DUMMY; tk.position = B; tk.start = C; tk.end = C + 3; tk.byte_len = 0;
                } else {
                    const uint32_t bs = cbyte[se.x];
                    tk.id = sid > 0 ? sid : -sid;
                    tk.cls = sid > 0 ? KGPU_CLASS_KNOWN : KGPU_CLASS_UNKNOWN;
                    tk.position = bs; tk.start = se.x; tk.end = se.y; tk.byte_len = cbyte[se.y] - bs;
                }
                a.stage[ts + k] = tk;
            }
        }
        if (lane == 0) { a.status[s] = KGPU_SENT_OK; a.tok_count[s] = K; }
        if (a.dump_lattice && lane == 0) {  // kgpu_lattice_dump: the host reads the lattice straight out of the two slabs
            a.ctl->dump[0] = (unsigned long long)(sa.ptr - a.arena); a.ctl->dump[1] = (unsigned long long)(sn.ptr - a.arena);
            a.ctl->dump[2] = B; a.ctl->dump[3] = C; a.ctl->dump[4] = N; a.ctl->dump[5] = 1;
        }
        if (a.count_work) {
            wT = wave_sum(wT);
            accW[0] += 1; accW[1] += B; accW[2] += C; accW[3] += wT; accW[4] += N - 1; accW[5] += bcast32(wE); accW[6] += K;
        }
        __syncthreads();
    }
    if (a.count_work && lane == 0)
        for (int k = 0; k < 7; ++k) atomicAdd(&a.ctl->work[k], (unsigned long long)accW[k]);
}

// Exclusive scan of per-sentence token counts -> tok_offsets (single workgroup;
// n is at most a few million per batch).
// It is also the last reader of the launch's Control block: it publishes the block to the
// pinned host copy (no D2H copy node) and zeroes the device copy for the next launch (no memset
// node) -- two operations fewer per batch on a host-launch-bound stream.
// (launched with 1024 threads behind a pool-only chain, with 256 or 64 behind a chain that holds a windowed launch: there every SIMD is full of 128-VGPR
// wavefronts, and a workgroup of sixteen wavefronts waits until a whole CU has drained -- 180 us on the chain's critical path on cfg 5, round 5)
__global__ __launch_bounds__(1024) void k_scan_counts(BatchArgs a, Control *host_ctl) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < a.n; base += nthr) {
        const uint64_t i = base + tid;
        const uint32_t v = i < a.n ? a.tok_count[i] : 0;
        const uint32_t vs = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wid] = vs;
        __syncthreads();
        uint64_t woff = 0;
        for (uint32_t w = 0; w < wid; ++w) woff += wsum[w];
        const uint64_t carry = carry_s;
        if (i < a.n) a.tok_offsets[i] = carry + woff + vs - v;
        __syncthreads();
        if (tid == nthr - 1) carry_s = carry + woff + vs;
        __syncthreads();
    }
    if (tid == 0) { a.tok_offsets[a.n] = carry_s; if (a.toff8) a.toff8[a.n] = carry_s; }   // (the mirrored table's last entry here, not in the compaction: an empty shard has no sentence to write it)
    static_assert(sizeof(Control) % 4 == 0, "Control is copied dword by dword");
    for (uint32_t k = tid; k < sizeof(Control) / 4; k += nthr) {
        uint32_t *dc = (uint32_t *)a.ctl, *hc = (uint32_t *)host_ctl;
        uint32_t v = dc[k];
        const uint32_t nt = (uint32_t)(offsetof(Control, n_tokens) / 4);
        if (k == nt) v = (uint32_t)carry_s;
        if (k == nt + 1) v = (uint32_t)(carry_s >> 32);
        __hip_atomic_store(&hc[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dc[k] = 0;
    }
}

// Staging (dequeue order) -> dense sentence order.  One wavefront per sentence,
// dword-granular coalesced copy.
__global__ __launch_bounds__(256) void k_compact(BatchArgs a) {
    const uint32_t lane = threadIdx.x & 63, wpb = blockDim.x >> 6;   // (four wavefronts per workgroup, or one behind a chain with a windowed launch: see k_scan_counts)
    const uint64_t wave = (uint64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * wpb;
    for (uint64_t s = wave; s < a.n; s += nwaves) {
        const uint32_t cnt = a.tok_count[s];
        const uint64_t dst = a.tok_offsets[s];
        if (dst + cnt > a.out_cap) continue;
        const uint32_t *src = (const uint32_t *)(a.stage + (a.offsets[s] - a.offsets[0] + s));
        uint32_t *out = (uint32_t *)(a.out + dst);
        for (uint32_t w = lane; w < cnt * 6; w += 64) out[w] = src[w];
    }
}

// Staging -> dense sentence order as 8-byte records (kgpu_token8) + the first token's (position, start) per sentence.
// One wavefront per sentence, one 8-byte store per token: when a.out8 is pinned host memory these stores are the transfer.
// (runs behind k_scan_counts, which has already published and zeroed the control block: the packing-overflow flag goes straight to the host copy)
__global__ __launch_bounds__(256) void k_compact8(BatchArgs a, Control *host_ctl) {
    const uint32_t lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * wpb;
    for (uint64_t s = wave; s < a.n; s += nwaves) {
        const uint32_t cnt = a.tok_count[s];
        const uint64_t dst = a.tok_offsets[s];
        const kgpu_token *src = a.stage + (a.offsets[s] - a.offsets[0] + s);
        if (lane == 0) {
            const uint2 f = cnt ? make_uint2(src[0].position, src[0].start) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            *(uint2 *)(a.first8 + 2 * s) = f;
            if (a.status8) a.status8[s] = a.status[s];
            if (a.toff8) a.toff8[s] = dst;   // ([n] comes from k_scan_counts)
        }
        if (dst + cnt > a.out_cap) continue;
        bool bad = false;
        for (uint32_t k = lane; k < cnt; k += 64) {
            const kgpu_token t = src[k];
            const uint32_t chars = t.end - t.start;
            bad |= chars > 0xFFFu || t.byte_len > 0x3FFFFu;
            *(uint2 *)(a.out8 + dst + k) = make_uint2((uint32_t)t.id, t.cls | (chars << 2) | (t.byte_len << 14));
        }
        if (__ballot(bad) != 0 && lane == 0) __hip_atomic_store(&host_ctl->pack_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Scan + compaction in ONE launch (batches of up to 65 536 sentences): workgroup w owns `per_wg` consecutive sentences; it sums the token counts in
// front of them itself (a few KB of L2 reads) instead of waiting for a scan kernel -- whose single 1024-thread workgroup had to find a CU with
// sixteen free wavefront slots on a chip full of pool-kernel wavefronts (8.5 us per batch, a second launch on the stream).  The LAST workgroup knows
// the total: it writes tok_offsets[n], publishes the control block to the host copy and zeroes the device copy (every field but pack_overflow, which
// this kernel's workgroups set in the host copy directly; the host clears it before the launch).
template <bool REC8>
__global__ __launch_bounds__(256) void k_scan_compact(BatchArgs a, Control *host_ctl, uint32_t per_wg) {
    __shared__ uint64_t wsum[4];
    __shared__ uint64_t loff[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t s0 = (uint64_t)blockIdx.x * per_wg;
    uint64_t part = 0;
    for (uint64_t i = tid; i < s0; i += 256) part += a.tok_count[i];
    part = wave_sum64(part);
    if (lane == 0) wsum[wid] = part;
    __syncthreads();
    const uint64_t prefix = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const uint64_t si = s0 + tid;
    const uint32_t v = (tid < per_wg && si < a.n) ? a.tok_count[si] : 0u;
    const uint32_t vs = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wid] = vs;
    __syncthreads();
    uint64_t woff = 0;
    for (uint32_t w = 0; w < wid; ++w) woff += wsum[w];
    const uint64_t mine = prefix + woff + vs - v;
    loff[tid] = mine;
    if (tid < per_wg && si < a.n) a.tok_offsets[si] = mine;
    const bool last_wg = s0 + per_wg >= a.n;
    const uint64_t total = prefix + wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (last_wg) {
        if (tid == 0) { a.tok_offsets[a.n] = total; if (a.toff8) a.toff8[a.n] = total; }   // (n = 0 included: an empty shard's mirrored table is {0})
        static_assert(sizeof(Control) % 4 == 0 && sizeof(Control) / 4 <= 256 * 4, "Control is copied a few dwords per thread");
        for (uint32_t k = tid; k < sizeof(Control) / 4; k += 256) {
            uint32_t *dc = (uint32_t *)a.ctl, *hc = (uint32_t *)host_ctl;
            uint32_t x = dc[k];
            const uint32_t nt = (uint32_t)(offsetof(Control, n_tokens) / 4);
            if (k == nt) x = (uint32_t)total;
            if (k == nt + 1) x = (uint32_t)(total >> 32);
            if (k != (uint32_t)(offsetof(Control, pack_overflow) / 4)) __hip_atomic_store(&hc[k], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            dc[k] = 0;
        }
    }
    __syncthreads();
    for (uint32_t t = wid; t < per_wg; t += 4) {   // one wavefront per sentence
        const uint64_t sx = s0 + t;
        if (sx >= a.n) break;
        const uint64_t dst = loff[t];
        const uint64_t nxt = t + 1 < 256 ? loff[t + 1] : total;  // (threads past the workgroup's sentences scanned zeros)
        const uint32_t cnt = (uint32_t)(nxt - dst);
        const kgpu_token *src = a.stage + (a.offsets[sx] - a.offsets[0] + sx);
        if constexpr (REC8) {
            if (lane == 0) {
                const uint2 f = cnt ? make_uint2(src[0].position, src[0].start) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                *(uint2 *)(a.first8 + 2 * sx) = f;
                if (a.status8) a.status8[sx] = a.status[sx];
                if (a.toff8) a.toff8[sx] = dst;
            }
            if (dst + cnt > a.out_cap) continue;
            bool bad = false;
            for (uint32_t k = lane; k < cnt; k += 64) {
                const kgpu_token tk = src[k];
                const uint32_t chars = tk.end - tk.start;
                bad |= chars > 0xFFFu || tk.byte_len > 0x3FFFFu;
                *(uint2 *)(a.out8 + dst + k) = make_uint2((uint32_t)tk.id, tk.cls | (chars << 2) | (tk.byte_len << 14));
            }
            if (__ballot(bad) != 0 && lane == 0) __hip_atomic_store(&host_ctl->pack_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            if (dst + cnt > a.out_cap) continue;
            const uint32_t *srcw = (const uint32_t *)src;
            uint32_t *out = (uint32_t *)(a.out + dst);
            for (uint32_t w = lane; w < cnt * 6; w += 64) out[w] = srcw[w];
        }
    }
}

int launch_tokenize_pool(const DictView &d, const BatchArgs &a, const WorkIO &io, uint32_t pool_bytes, uint32_t waves,
                         uint32_t max_pages, int n_workgroups, uint32_t stop_after, void *stream);  // kgpu_pool.hip
int pool_workgroups_per_cu(uint32_t pool_bytes, uint32_t waves);

// Launch chain: the LDS page-pool kernel(s) -> the windowed kernel (whatever the pools route away: long sentences, lattices too dense for a
// pool) -> the general kernel (what the windowed kernel hands back: the last resort).  Every launch is a persistent grid over its work list
// (the first one: the identity over [0, n)) and pushes what it does not serve onto the next launch's list.
int launch_tokenize_window(const DictView &d, const BatchArgs &a, const WorkIO &io, uint32_t lds_bytes, int n_workgroups, int team, void *stream, bool claim = false);  // kgpu_window.hip

static int launch_window_over(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, const uint32_t *in_list, const unsigned int *in_count, int li, void *stream, int grid = 0) {
    WorkIO io{in_list, in_count, a.ovf[li], &a.ctl->ovf_count[li], nullptr};
    uint64_t wg = plan.window_workgroups;
    if (!in_list && a.n < wg) wg = a.n;
    // behind the pools the list's length is on the device; a grid of the chip's full size is mostly workgroups that find nothing -- and each of them has to find a free
    // slot on a chip full of long-running wavefronts before it can say so, which is what the launch (and the scan behind it) then waits for: the host's estimate instead
    if (in_list && grid > 0 && (uint64_t)grid < wg) wg = (uint64_t)grid;
    // more sentences expected than workgroups: they are claimed one by one instead of every G-th being a workgroup's (kgpu_window.hip; KGPU_WINDOW_CLAIM=0 / 1 forces it)
    static const int claim_mode = [] { const char *e = getenv("KGPU_WINDOW_CLAIM"); return e ? atoi(e) : -1; }();
    const uint64_t expected = in_list ? (grid > 64 ? ((uint64_t)grid - 64) * 4 / 5 : 0) : a.n;
    const bool claim = claim_mode >= 0 ? claim_mode != 0 : expected > wg;
    return launch_tokenize_window(d, a, io, plan.window_lds_bytes, (int)(wg ? wg : 1), 1, stream, claim);
}
static int launch_general_over(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, const uint32_t *in_list, const unsigned int *in_count, uint32_t stop_after, void *stream) {
    WorkIO io{in_list, in_count, nullptr, nullptr, nullptr};
    uint64_t wg = plan.general_workgroups;
    if (!in_list && a.n < wg) wg = a.n;
    hipLaunchKernelGGL(k_tokenize_general, dim3((unsigned)(wg ? wg : 1)), dim3(64), 0, (hipStream_t)stream, d, a, io, stop_after);
    return (int)hipGetLastError();
}

int launch_tokenize(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, int n_pools_now, uint32_t stop_after, void *stream,
                    void *event_after_first, bool window_now, bool tail_now, bool team_now, int window_grid) {
    Control *ctl = a.ctl;
    const uint32_t *in_list = nullptr;
    const unsigned int *in_count = nullptr;
    int li = 0;  // next free work list
    for (int k = 0; k < plan.n_pools && k < n_pools_now; ++k, ++li) {
        WorkIO io{in_list, in_count, a.ovf[li], &ctl->ovf_count[li], &ctl->late_count[li]};
        uint64_t wg = plan.pool_workgroups[k];
        const uint64_t want = (a.n + plan.pool_waves[k] - 1) / plan.pool_waves[k];
        if (!in_list && want < wg) wg = want;
        int e = launch_tokenize_pool(d, a, io, plan.pool_bytes[k], plan.pool_waves[k], plan.pool_max_pages[k], (int)(wg ? wg : 1), stop_after, stream);
        if (e) return e;
        if (k == 0 && event_after_first && hipEventRecord((hipEvent_t)event_after_first, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
        in_list = a.ovf[li];
        in_count = &ctl->ovf_count[li];
    }
    if (event_after_first && (plan.n_pools == 0 || n_pools_now == 0) && hipEventRecord((hipEvent_t)event_after_first, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
    if (window_now && plan.window_lds_bytes && stop_after == 0) {
        bool behind_team = false;
        if (team_now && !in_list && plan.window_team_workgroups > 0 && a.n) {
            // a short list of long sentences: two wavefronts per sentence (kgpu_window.hip, TEAM), one workgroup each; what that form cannot hold goes on to
            // the ordinary form behind it
            WorkIO io{nullptr, nullptr, a.ovf[li], &ctl->ovf_count[li], nullptr};
            int e = launch_tokenize_window(d, a, io, plan.window_lds_bytes, (int)std::min<uint64_t>(a.n, 1u << 30), plan.window_team, stream);
            if (e) return e;
            in_list = a.ovf[li];
            in_count = &ctl->ovf_count[li];
            ++li;
            behind_team = true;
        }
        int e = launch_window_over(d, a, plan, in_list, in_count, li, stream, behind_team ? 256 : window_grid);   // (what a team hands on is rare: a small strided grid)
        if (e) return e;
        in_list = a.ovf[li];
        in_count = &ctl->ovf_count[li];
        ++li;
    }
    // No recent batch left a sentence for the rest of the chain: it is left out (an empty launch still costs its 5.5 us on the stream, 4 % of a
    // cfg 2 batch's chain -- and far more behind a chip full of long-running wavefronts); the host finds a sentence that needed it in the last
    // list's count and launches what is missing over that list (launch_tail_only).
    if (!tail_now && in_list) return (int)hipGetLastError();
    return launch_general_over(d, a, plan, in_list, in_count, stop_after, stream);
}

// What a chain that ended on work list `li` left out (kgpu_api.cpp: enqueue_tail): the windowed kernel over that list unless it was in the chain,
// then the general kernel over what is left.
int launch_tail_only(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, int li, bool window_was_in_chain, void *stream) {
    Control *ctl = a.ctl;
    if (li < 0 || li > 2) return (int)hipErrorInvalidValue;
    if (!window_was_in_chain && plan.window_lds_bytes) {
        int e = launch_window_over(d, a, plan, a.ovf[li], &ctl->ovf_count[li], li + 1, stream);
        if (e) return e;
        ++li;
    }
    return launch_general_over(d, a, plan, a.ovf[li], &ctl->ovf_count[li], 0u, stream);
}

int launch_small_call(const DictView &d, const BatchArgs &a, const LaunchPlan &plan, void *stream) {
    Control *ctl = a.ctl;
    WorkIO io{nullptr, nullptr, a.ovf[0], &ctl->ovf_count[0], &ctl->late_count[0]};
    // every wavefront gets at most one sentence -- and a workgroup of its own with a 64 KB LDS slice (at most 128 of them: the chip has room): no pages to
    // reserve, no wavefronts that start only to find nothing and meet at the barrier (one sentence: 52 -> 44.5 us per call, 64: 65.7 -> 57.3; round 6),
    // and anything up to ~350 characters stays in this one launch.  KGPU_SMALL_POOL=<KiB>:<wavefronts> (measurement) overrides.
    static const struct Shape { uint32_t bytes, waves; } sh = [] {
        Shape s{64u * 1024u, 1u};
        if (const char *e = getenv("KGPU_SMALL_POOL")) { int k = atoi(e), w = 1; if (const char *c = strchr(e, ':')) w = atoi(c + 1); if (k >= 8 && k <= 64 && w >= 1 && w <= 16) s = Shape{(uint32_t)k * 1024u, (uint32_t)w}; }
        return s;
    }();
    const uint64_t wg = (a.n + sh.waves - 1) / sh.waves;
    return launch_tokenize_pool(d, a, io, sh.bytes, sh.waves, 64u, (int)(wg ? wg : 1), 0u, stream);
}

int launch_general_only(const DictView &d, const BatchArgs &a, void *stream) {
    WorkIO io{nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_tokenize_general, dim3(1), dim3(64), 0, (hipStream_t)stream, d, a, io, 0u);
    return (int)hipGetLastError();
}

int launch_scan_compact(const BatchArgs &a, Control *host_ctl, void *stream, bool small_workgroups, bool small_scan_only) {
    // Measured (tools/ab_scan.sh): records bound for mapped host memory (the large host call: a.toff8) 83.8 against 66.6 M sentences/s end to end in one
    // launch; the device-resident 24-byte path 92.8 against 96.9 -- there the separate kernels stay.  KGPU_SCAN_COMPACT=1 / 2 force one form (experiments),
    // KGPU_SCAN_WG the sentences per workgroup.
    static const int mode = [] { const char *e = getenv("KGPU_SCAN_COMPACT"); return e ? atoi(e) : 0; }();
    static const int wg_env = [] { const char *e = getenv("KGPU_SCAN_WG"); const int v = e ? atoi(e) : 0; return (v >= 4 && v <= 256) ? v : 0; }();
    if (a.n <= 65536 && (mode == 1 || (mode == 0 && a.toff8))) {
        uint32_t per_wg = wg_env ? (uint32_t)wg_env : a.n <= 4096 ? 64u : a.n <= 16384 ? 128u : 256u;
        if ((a.n + per_wg - 1) / per_wg > 65536) per_wg = 256;
        const uint64_t wgs = a.n ? (a.n + per_wg - 1) / per_wg : 1;
        if (a.out8) hipLaunchKernelGGL(k_scan_compact<true>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a, host_ctl, per_wg);
        else hipLaunchKernelGGL(k_scan_compact<false>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a, host_ctl, per_wg);
        return (int)hipGetLastError();
    }
    // small_workgroups: behind a chain that holds a windowed launch the chip is full of single-wavefront workgroups at four 128-VGPR wavefronts per SIMD -- a
    // workgroup of one wavefront finds a place as soon as ANY of them ends, one of sixteen (or four) needs a CU (or a SIMD row) to drain
    const unsigned scan_threads = !(small_workgroups || small_scan_only) ? 1024u : a.n > 1024 ? 256u : 64u, wpb = small_workgroups ? 1u : 4u;
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(scan_threads), 0, (hipStream_t)stream, a, host_ctl);
    uint64_t blocks = (a.n + wpb - 1) / wpb;
    if (blocks > 2048 * (4 / wpb)) blocks = 2048 * (4 / wpb);
    if (blocks == 0) blocks = 1;
    // (measured: the compaction of 8-byte records into mapped host memory on a stream of its own, behind an event -- so that the context's
    // stream goes on with the next batch -- gives 46.8 instead of 68.6 M sentences/s end to end: one more stream than hardware queues)
    if (a.out8) hipLaunchKernelGGL(k_compact8, dim3((unsigned)blocks), dim3(64 * wpb), 0, (hipStream_t)stream, a, host_ctl);
    else hipLaunchKernelGGL(k_compact, dim3((unsigned)blocks), dim3(64 * wpb), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

LaunchPlan default_launch_plan(int device) {
    hipDeviceProp_t p;
    int cus = 256;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) cus = p.multiProcessorCount;
    LaunchPlan t{};
    t.general_workgroups = cus * 2;  // the last resort is rarely needed: few workgroups, so that an empty launch drains quickly on a busy chip
    // Default: four 40 KB pools per CU with 4 wavefronts each (16 sentences in flight per CU, any mix of
    // sizes).  A workgroup holds its LDS until its last wavefront is through, and the next launch's workgroups
    // start only then: four-wavefront workgroups drain sooner at the tail of a 4096-sentence batch than eight-
    // or sixteen-wavefront ones (80:8 66.3, 160:16 63.9, 40:4 68.6 M sentences/s; two-wavefront pools lose to
    // fragmentation, and any shape that is not 16 wavefronts per CU loses to the batch size: 4096 = 256 x 16).
    // A sentence expected to need more than 40 of a pool's 64 pages (25 KB, ~155 chars) goes to the
    // long-sentence kernel instead: LDS x time grows with the square of the length, and a few long sentences
    // would otherwise hold the pools while the short ones wait (cfg 3 in batches of 16384, M sentences/s by this limit:
    // 16 pages 16.3, 24 16.8, 32 17.6-18.4, 40 17.8-18.8, 48 17.1-18.4; round 2, batches of 4096: 56 11.6, 64 10.7 against 12.3; cfg 2 is
    // indifferent: 96.8-97.2 at 40, 96.4-97.0 at 48).
    // KGPU_POOL="<KiB>:<wavefronts>[:<max pages>][,...]", "0" = none.
    t.n_pools = 0;
    {
        const char *e = getenv("KGPU_POOL");
        const char *q = e ? e : "40:4:32";
        t.pool_limit_auto = e == nullptr;
        while (*q && t.n_pools < 2) {
            int kib = atoi(q), w = 8, mp = 64;
            const char *c = q;
            while (*c && *c != ',' && *c != ':') ++c;
            if (*c == ':' && atoi(c + 1) > 0) {
                w = atoi(c + 1);
                ++c;
                while (*c && *c != ',' && *c != ':') ++c;
                if (*c == ':' && atoi(c + 1) > 0) mp = atoi(c + 1);
            }
            if (w > 16) w = 16;
            if (mp > 64) mp = 64;
            if (kib >= 8 && kib <= 160) {
                const int per_cu = pool_workgroups_per_cu((uint32_t)kib * 1024, (uint32_t)w);
                if (per_cu > 0) {
                    t.pool_bytes[t.n_pools] = (uint32_t)kib * 1024; t.pool_waves[t.n_pools] = (uint32_t)w;
                    t.pool_max_pages[t.n_pools] = (uint32_t)mp;
                    t.pool_workgroups[t.n_pools] = cus * per_cu;
                    ++t.n_pools;
                }
            }
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    t.alt_pool_bytes = 20 * 1024; t.alt_pool_waves = 2;
    {
        const int per_cu = (t.pool_limit_auto && t.n_pools) ? pool_workgroups_per_cu(t.alt_pool_bytes, t.alt_pool_waves) : 0;
        t.alt_pool_workgroups = per_cu > 0 ? cus * per_cu : 0;
    }
    // windowed kernel (everything the pools route away): KGPU_WINDOW="<KiB>" of LDS per single-wavefront workgroup, "0" = off (the general kernel then serves it all)
    {
        const char *e = getenv("KGPU_WINDOW");
        int kib = e ? atoi(e) : 10;   // 10 KB: 16 workgroups per CU = the four wavefronts per SIMD its 128 VGPRs allow (round 4, after big buckets lost their pair tables: cfg 3 22.5 M sentences/s against 19.2 at 12 KB and 20.8 at 11, cfg 5 2.67 against 2.71 Gchar/s)
        if (kib < 8 || kib > 160) kib = 0;
        t.window_lds_bytes = (uint32_t)kib * 1024;
        const int per_cu = kib ? window_workgroups_per_cu(t.window_lds_bytes) : 0;
        t.window_workgroups = cus * per_cu;
        if (per_cu <= 0) t.window_lds_bytes = 0;
        const char *ts = getenv("KGPU_WINDOW_TEAM_SIZE");
        t.window_team = ts && atoi(ts) >= 2 && atoi(ts) <= 4 ? atoi(ts) : 2;
        t.window_team_workgroups = t.window_lds_bytes ? cus * std::max(0, window_team_workgroups_per_cu(t.window_lds_bytes, t.window_team)) : 0;
        const char *tm = getenv("KGPU_WINDOW_TEAM");
        t.window_team_mode = tm ? atoi(tm) : -1;
        const char *wf = getenv("KGPU_WINDOW_FIRST");
        t.window_first_bytes = (uint32_t)std::max(0, wf ? atoi(wf) : 1024);
    }
    if (const char *e = getenv("KGPU_GENERAL_WG")) { int v = atoi(e); if (v > 0) t.general_workgroups = v; }
    if (const char *e = getenv("KGPU_POOL_WG")) { int v = atoi(e); if (v > 0 && t.n_pools) t.pool_workgroups[0] = v; }
    return t;
}

}  // namespace kgpu

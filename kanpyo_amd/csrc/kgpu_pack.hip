// kanpyo_amd/csrc/kgpu_pack.hip -- packed LDS-resident tokenize kernel (gfx950).
//
// Same algorithm as kgpu_lds.hip, but one 64-lane wavefront owns a PACK of up to
// four consecutive short sentences at once.  The sentences are laid side by side
// in one global position space (sentence s occupies positions g_s .. g_s + C_s,
// the last one being its EOS), so the position-parallel phases (decode, trie
// walk, emit, connection-cost gather) simply see ~160 positions instead of ~40
// and fill the 64 lanes; and the Viterbi sweep, whose cost per step is a fixed
// chain of LDS round trips and scalar bookkeeping rather than arithmetic,
// advances all sentences of the pack in the SAME step: lane group g sweeps
// position r of sentence g.  One instruction stream therefore serves four
// sentences; the serial chain is max(C_s) steps long instead of sum(C_s).
//
// No word crosses a sentence boundary (walks stop at the sentence's last byte,
// unknown-word runs at its last char), each sentence has its own BOS bucket
// entry and EOS node, so the lattices stay independent and the result is
// bit-identical to running the sentences one at a time.  Node indices are global
// to the pack and ascending in (sentence, start position, insertion order), so
// the (total, node index) tie-break of lattice.rs:125-139 is unchanged.
//
// A pack that does not fit the tier's LDS, holds invalid UTF-8, or has more than
// MAXM dictionary prefixes at one position is handed, sentence by sentence, to
// the per-sentence tier chain (kgpu_lds.hip / kgpu_kernels.hip).
#include <cstdlib>

#include "kgpu_device.h"

namespace kgpu {

using namespace dev;

namespace {

constexpr uint32_t GMAX = 4;         // sentences per pack
constexpr uint32_t MAXM = 8;         // trie matches buffered per start position
constexpr uint32_t NONE16 = 0xFFFFu;
constexpr uint32_t BOSMARK = 0xFFFEu;  // predecessor "node" meaning the sentence's BOS

__device__ __forceinline__ uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) & ~(a - 1); }

}  // namespace

__global__ __launch_bounds__(64) void k_tokenize_pack(DictView d, BatchArgs a, TierIO io, uint32_t lds_bytes, uint32_t gpack) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    const int32_t base_root = d.da[1].base;
    const uint32_t lgshift = gpack == 4 ? 4 : gpack == 2 ? 5 : 6;  // lanes per sentence in the sweep = 1 << lgshift
    uint64_t accW[7] = {0, 0, 0, 0, 0, 0, 0};  // work counters of this workgroup, flushed once at exit

    const uint64_t npacks = (a.n + gpack - 1) / gpack;
    for (uint64_t pk = blockIdx.x; pk < npacks; pk += gridDim.x) {
        const uint64_t s0 = pk * gpack;
        const uint32_t G = (uint32_t)min((uint64_t)gpack, a.n - s0);
        // ---- sentence table (wave-uniform) ---------------------------------------
        uint32_t sByte[GMAX + 1];  // byte offset of each sentence inside the pack's text
        const uint64_t o0 = a.offsets[s0];
#pragma unroll
        for (uint32_t s = 0; s <= GMAX; ++s) sByte[s] = (uint32_t)(a.offsets[s0 + min(s, G)] - o0);
        const uint32_t Bt = sByte[GMAX];
        bool defer = (uint64_t)Bt + 128 > lds_bytes || Bt > 0xFFF0u || (((uint64_t)Bt * a.est_q8) >> 8) + 1024 > lds_bytes;
        uint32_t P = 0, N = 0, Nb = 0, E = 0, C = 0, wT = 0;
        uint32_t sGpos[GMAX + 1], sC[GMAX];
#pragma unroll
        for (uint32_t s = 0; s < GMAX; ++s) { sGpos[s] = 0; sC[s] = 0; }
        sGpos[GMAX] = 0;

        // LDS pointers (carved below)
        uint8_t *text = smem;
        uint32_t *nb = nullptr, *boff = nullptr, *bfill = nullptr, *ebase = nullptr, *mid = nullptr;
        uint16_t *cbyte = nullptr, *uspan = nullptr, *path = nullptr, *cp16 = nullptr, *send = nullptr;
        uint8_t *ccat = nullptr, *mcnt = nullptr, *mnch = nullptr;
        uint16_t *stab = nullptr;  // per sentence: {gpos0, C, byte0, K}
        uint32_t off = 0, moff = 0;
        const uint8_t *gtext = a.utf8 + o0;

        if (!defer) {
            // ---- phase 0a: stage the pack's bytes in LDS, count chars per sentence --------
            uint32_t cs[GMAX + 1] = {0, 0, 0, 0, 0};  // chars before each sentence start
            for (uint32_t k0 = 0; k0 < Bt + 4; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t b = k < Bt ? gtext[k] : 0x80u;
                if (k < Bt + 4) text[k] = (uint8_t)b;
                const uint64_t m = __ballot(k < Bt && (b & 0xC0) != 0x80);
#pragma unroll
                for (uint32_t s = 1; s <= GMAX; ++s) {
                    const uint32_t x = sByte[s];
                    if (x >= k0 + 64) cs[s] += __popcll(m);
                    else if (x > k0) cs[s] += __popcll(m & ((1ull << (x - k0)) - 1));
                }
            }
            C = cs[GMAX];
#pragma unroll
            for (uint32_t s = 0; s < GMAX; ++s) { sC[s] = cs[s + 1] - cs[s]; sGpos[s] = cs[s] + s; }
            sGpos[GMAX] = C + G;
            P = C + G;  // one extra position per sentence: its EOS
            // ---- LDS carve ----------------------------------------------------------------
            off = align_up(Bt + 4, 4);
            stab = (uint16_t *)(smem + off);  off += 2 * 4 * GMAX;
            nb = (uint32_t *)(smem + off);    off += 4 * (P + 2);
            boff = (uint32_t *)(smem + off);  off += 4 * (P + 2);
            bfill = (uint32_t *)(smem + off); off += 4 * (P + 2);
            ebase = (uint32_t *)(smem + off); off += 4 * (P + 2);
            cbyte = (uint16_t *)(smem + off); off += 2 * (P + 2);
            uspan = (uint16_t *)(smem + off); off += 2 * (P + 2);
            path = (uint16_t *)(smem + off);  off += 2 * (P + 2);
            cp16 = (uint16_t *)(smem + off);  off += 2 * (P + 2);
            send = (uint16_t *)(smem + off);  off += 2 * (P + 2);
            ccat = smem + off;                off += align_up(P + 2, 4);
            mcnt = smem + off;                off += align_up(P + 2, 4);
            const uint32_t mbytes = align_up(P * MAXM * 5, 16);
            if (off + mbytes > lds_bytes) defer = true;
            moff = (lds_bytes - mbytes) & ~15u;
            mid = (uint32_t *)(smem + moff);
            mnch = smem + moff + 4 * P * MAXM;
        }
        __syncthreads();

        if (!defer) {
            // ---- phase 0b: decode + validate + category, positions are global to the pack ----
            uint32_t cb = 0, bad = 0, lensum = 0;
            for (uint32_t k0 = 0; k0 < Bt; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t b = k < Bt ? text[k] : 0x80u;
                const bool start = k < Bt && (b & 0xC0) != 0x80;
                const uint64_t m = __ballot(start);
                const uint32_t ci = cb + __popcll(m & ((1ull << lane) - 1));
                if (start) {
                    uint32_t sidx = 0;
#pragma unroll
                    for (uint32_t s = 1; s < GMAX; ++s) sidx += (s < G && k >= sByte[s]) ? 1u : 0u;
                    uint32_t send_byte = sByte[1], g0 = sGpos[0], cS = sC[0];
#pragma unroll
                    for (uint32_t s = 1; s < GMAX; ++s) if (sidx == s) { send_byte = sByte[s + 1]; g0 = sGpos[s]; cS = sC[s]; }
                    uint32_t l, cp;
                    if (b < 0x80) { l = 1; cp = b; }
                    else if (b >= 0xC2 && b <= 0xDF) { l = 2; cp = b & 0x1F; }
                    else if ((b & 0xF0) == 0xE0) { l = 3; cp = b & 0x0F; }
                    else if (b >= 0xF0 && b <= 0xF4) { l = 4; cp = b & 0x07; }
                    else { l = 1; cp = 0; bad = 1; }
                    if (k + l > send_byte) { bad = 1; l = 1; }  // a char may not run past its sentence
                    for (uint32_t j = 1; j < l; ++j) {
                        const uint32_t bb = text[k + j];
                        if ((bb & 0xC0) != 0x80) bad = 1;
                        cp = (cp << 6) | (bb & 0x3F);
                    }
                    if (l == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) bad = 1;
                    if (l == 4 && (cp < 0x10000 || cp > 0x10FFFF)) bad = 1;
                    lensum += l;
                    const uint32_t pos = ci + sidx;
                    cbyte[pos] = (uint16_t)k;
                    cp16[pos] = (uint16_t)(cp < 0xFFFFu ? cp : 0xFFFFu);
                    ccat[pos] = bad ? 0 : (cp < d.cat_len ? d.cat[cp] : d.cat[0]);  // char_category_def.rs:33-38
                    send[pos] = (uint16_t)(g0 + cS);
                }
                cb += __popcll(m);
            }
            lensum = bcast32(wave_sum(lensum));
            if (__ballot(bad != 0) != 0 || lensum != Bt) defer = true;  // invalid UTF-8: the per-sentence path flags it
            if (lane < G) {  // EOS position of sentence `lane`
                uint32_t g0 = sGpos[0], cS = sC[0], be = sByte[1], bs = sByte[0];
#pragma unroll
                for (uint32_t s = 1; s < GMAX; ++s) if (lane == s) { g0 = sGpos[s]; cS = sC[s]; be = sByte[s + 1]; bs = sByte[s]; }
                const uint32_t pe = g0 + cS;
                cbyte[pe] = (uint16_t)be; send[pe] = (uint16_t)pe; ccat[pe] = 0; cp16[pe] = 0xFFFFu;
                stab[lane * 4 + 0] = (uint16_t)g0; stab[lane * 4 + 1] = (uint16_t)cS; stab[lane * 4 + 2] = (uint16_t)bs;
            }
            for (uint32_t e = lane; e < P + 2; e += 64) { boff[e] = 0; bfill[e] = 0; }
        }
        __syncthreads();

        uint32_t maxpairs = 0;
        if (!defer) {
            // ---- phase 1: one trie walk per char position; count + park matches ---------------
            uint32_t ovf = 0;
            const int nchunks = (int)((P + 63) / 64);
            uint32_t carry_end = P;
            for (int ch = nchunks - 1; ch >= 0; --ch) {
                const uint32_t i = (uint32_t)ch * 64 + lane;
                const bool inr = i < P;
                const uint32_t se = inr ? send[i] : 0;
                const bool active = inr && i != se;  // a char position (not an EOS slot)
                const uint32_t cat = active ? ccat[i] : 0x1FFu;
                // a run of one category ends at a category change or at the sentence's last char
                const bool brk = inr && (!active || i + 1 == se || ccat[i + 1] != cat);
                const uint64_t bm = __ballot(brk);
                const uint64_t rest = bm >> lane;
                const uint32_t run_end = rest ? i + (uint32_t)__ffsll((unsigned long long)rest) : carry_end;
                carry_end = bcast32(run_end);
                if (inr && !active) { nb[i] = 1; uspan[i] = 0; mcnt[i] = 0; }  // the EOS node starts here (lattice.rs:165-175)
                if (active) {
                    uint32_t cnt = 0, m = 0;
                    const uint32_t bend = cbyte[se];  // last byte (exclusive) of this sentence
                    auto on_match = [&](uint32_t id, uint32_t nch) {
                        if (m < MAXM && nch < 256) { mid[i * MAXM + m] = id; mnch[i * MAXM + m] = (uint8_t)nch; }
                        else ovf = 1;
                        ++m;
                        const uint32_t nrec = 1u + d.morph[id - 1].dup;  // index.rs:46-51
                        cnt += nrec;
                        atomicAdd(&boff[i + nch], nrec);
                    };
                    const uint32_t cp = cp16[i];
                    if (cp == 0xFFFFu) {
                        wT += da_walk(d, text, cbyte[i], bend, base_root, on_match);
                    } else {
                        const DaNode f = d.first[cp];  // {.base = node, .check = base[node]} or {0, steps}
                        if (f.base == 0) {
                            wT += (uint32_t)f.check;
                        } else {
                            int32_t p = f.base, bp = f.check;
                            uint32_t k = cbyte[i + 1], nstart = 1;
                            wT += k - cbyte[i];
                            for (;;) {
                                const bool more = k < bend;
                                const uint32_t c = more ? text[k] : 0u;
                                const bool boundary = !more || (c & 0xC0) != 0x80;
                                const uint32_t q = (uint32_t)(bp + (int32_t)c);
                                const bool doprobe = boundary && (uint32_t)bp < d.da_len;
                                const bool donext = more && q < d.da_len;
                                DaNode t{0, 0}, nx{0, 0};
                                if (doprobe) t = d.da[bp];  // + TERMINATOR (da.rs:166)
                                if (donext) nx = d.da[q];
                                if (doprobe && t.check == p && t.base < 0) on_match((uint32_t)(-t.base), nstart);
                                if (!more) break;
                                ++wT;
                                if (!donext || nx.check != p) break;  // da.rs:162-165
                                p = (int32_t)q;
                                bp = nx.base;
                                nstart += boundary;
                                ++k;
                            }
                        }
                    }
                    mcnt[i] = (uint8_t)(m < MAXM ? m : MAXM);
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
                    }
                    uspan[i] = (uint16_t)span;
                    nb[i] = cnt;
                }
            }
            if (__ballot(ovf != 0) != 0) defer = true;
            if (lane < G) atomicAdd(&boff[stab[lane * 4 + 0]], 1u);  // each sentence's BOS ends at its first position
            if (lane == 0) { nb[P] = 0; nb[P + 1] = 0; }
        }
        __syncthreads();

        uint16_t *pre = nullptr, *nLeft = nullptr, *nSlot = nullptr, *nStart = nullptr;
        int16_t *nCost = nullptr, *mpair = nullptr;
        int32_t *nSid = nullptr;
        uint2 *bk = nullptr;
        if (!defer) {
            // ---- phase 2: prefix sums ---------------------------------------------------------
            uint32_t ncarry = 0, bcarry = 0, ecarry = 0;
            for (uint32_t i0 = 0; i0 < P + 2; i0 += 64) {
                const uint32_t i = i0 + lane;
                const uint32_t v = i < P + 2 ? nb[i] : 0;
                const uint32_t w = i < P + 2 ? boff[i] : 0;
                const uint32_t x = v * w;
                const uint32_t vs = wave_incl_scan(v, lane), ws = wave_incl_scan(w, lane), xs = wave_incl_scan(x, lane);
                if (i < P + 2) { nb[i] = ncarry + vs - v; boff[i] = bcarry + ws - w; ebase[i] = ecarry + xs - x; }
                ncarry += __shfl(vs, 63, 64);
                bcarry += __shfl(ws, 63, 64);
                ecarry += __shfl(xs, 63, 64);
                maxpairs = max(maxpairs, x);
            }
            N = bcast32(ncarry); Nb = bcast32(bcarry); E = bcast32(ecarry);
            // ---- LDS carve, part 2 ---------------------------------------------------------------
            off = align_up(off, 8);
            bk = (uint2 *)(smem + off);          off += 8 * Nb;  // bucket (= edges[e]): {dp, right | node << 16}
            nSid = (int32_t *)(smem + off);      off += 4 * N;
            nLeft = (uint16_t *)(smem + off);    off += 2 * N;
            nCost = (int16_t *)(smem + off);     off += 2 * N;
            nSlot = (uint16_t *)(smem + off);    off += 2 * N;
            nStart = (uint16_t *)(smem + off);   off += 2 * N;
            off = align_up(off, 4);
            const uint32_t off_emit_end = off;
            pre = (uint16_t *)(smem + off);      off += align_up(2 * N, 4);
            mpair = (int16_t *)(smem + off);
            const uint32_t mcap = off < lds_bytes ? (lds_bytes - off) / 2 : 0;
            if (N >= BOSMARK || off_emit_end > moff || off > lds_bytes || mcap < E) {
                defer = true;
                if (lane == 0) atomicAdd(io.late_count, G);
            }
        }
        __syncthreads();

        if (defer) {  // hand the pack to the per-sentence tier chain
            if (lane == 0) {
                const unsigned int kk = atomicAdd(io.out_count, G);
                for (uint32_t s = 0; s < G; ++s) io.out_list[kk + s] = (uint32_t)(s0 + s);
            }
            continue;
        }

        // ---- phase 3: emit nodes from the parked matches ----------------------------------------
        for (uint32_t i = lane; i < P; i += 64) {
            uint32_t t = nb[i];
            if (i == send[i]) {  // EOS: Morph(0,0,0), id 0, never a predecessor
                nLeft[t] = (uint16_t)d.eos_left; nCost[t] = 0; nSlot[t] = NONE16; nStart[t] = (uint16_t)i; nSid[t] = 0;
                continue;
            }
            const uint32_t nm = mcnt[i];
            for (uint32_t m = 0; m < nm; ++m) {
                const uint32_t id = mid[i * MAXM + m];
                const uint32_t end = i + mnch[i * MAXM + m];
                const uint32_t nrec = 1u + d.morph[id - 1].dup;
                for (uint32_t r = 0; r < nrec; ++r) {  // lattice.rs:177-188
                    const Morph8 mm = d.morph[id - 1 + r];
                    const uint32_t slot = boff[end] + atomicAdd(&bfill[end], 1u);
                    nLeft[t] = (uint16_t)mm.left; nCost[t] = mm.cost; nSlot[t] = (uint16_t)slot; nStart[t] = (uint16_t)i;
                    nSid[t] = (int32_t)(id + r);
                    bk[slot].y = (uint32_t)(uint16_t)mm.right | (t << 16);
                    ++t;
                }
            }
            const uint32_t span = uspan[i];
            if (span) {  // lattice.rs:87-97,190-201
                const CatInfo ci = d.cinfo[ccat[i]];
                const uint32_t end = i + span;
                for (uint32_t r = 0; r < ci.unk_count; ++r) {
                    const Morph8 mm = d.unk_morph[ci.unk_first - 1 + (int32_t)r];
                    const uint32_t slot = boff[end] + atomicAdd(&bfill[end], 1u);
                    nLeft[t] = (uint16_t)mm.left; nCost[t] = mm.cost; nSlot[t] = (uint16_t)slot; nStart[t] = (uint16_t)i;
                    nSid[t] = -(ci.unk_first + (int32_t)r);
                    bk[slot].y = (uint32_t)(uint16_t)mm.right | (t << 16);
                    ++t;
                }
            }
        }
        if (lane < G) {  // BOS of sentence `lane`: dp None -> 0 (lattice.rs:127), right_id 0
            const uint32_t g0 = stab[lane * 4 + 0];
            const uint32_t slot = boff[g0] + atomicAdd(&bfill[g0], 1u);
            bk[slot] = make_uint2(0u, (BOSMARK << 16) | d.bos_right);
        }
        __syncthreads();

        // ---- phase 3b: gather every connection cost into the LDS pair table (connection.rs:12-14)
        for (uint32_t t = lane; t < N; t += 64) {
            const uint32_t q = nStart[t];
            const uint32_t p0 = boff[q], Pq = boff[q + 1] - p0;
            const uint32_t tq0 = nb[q], T = nb[q + 1] - tq0, ti = t - tq0;
            const uint32_t base = ebase[q] + ti;  // pair (ti, j) at j*T + ti: lanes of one position read consecutive i16
            const int16_t *col = d.conn + (size_t)d.conn_rows * nLeft[t];
            uint32_t j = 0;
            for (; j + 4 <= Pq; j += 4) {  // 4 independent gathers in flight per lane
                const uint32_t r0 = bk[p0 + j].y & 0xFFFFu, r1 = bk[p0 + j + 1].y & 0xFFFFu;
                const uint32_t r2 = bk[p0 + j + 2].y & 0xFFFFu, r3 = bk[p0 + j + 3].y & 0xFFFFu;
                const int16_t c0 = col[r0], c1 = col[r1], c2 = col[r2], c3 = col[r3];
                mpair[base + j * T] = c0; mpair[base + (j + 1) * T] = c1;
                mpair[base + (j + 2) * T] = c2; mpair[base + (j + 3) * T] = c3;
            }
            for (; j < Pq; ++j) mpair[base + j * T] = col[bk[p0 + j].y & 0xFFFFu];
        }
        __syncthreads();

        // ---- phase 4: Viterbi sweep (lattice.rs:116-142): lane group g advances sentence g ------
        {
            const uint32_t g = lane >> lgshift, l = lane & ((1u << lgshift) - 1), LG = 1u << lgshift;
            const bool gv = g < G;
            const uint32_t qbase = gv ? stab[g * 4 + 0] : 0, Cg = gv ? stab[g * 4 + 1] : 0;
            uint32_t maxC = 0;
#pragma unroll
            for (uint32_t s = 0; s < GMAX; ++s) maxC = max(maxC, sC[s]);
            for (uint32_t r = 0; r <= maxC; ++r) {
                if (gv && r <= Cg) {
                    const uint32_t q = qbase + r;
                    const uint32_t t0 = nb[q], T = nb[q + 1] - t0;
                    const uint32_t p0 = boff[q], Pq = boff[q + 1] - p0;
                    const uint32_t eb = ebase[q];
                    for (uint32_t ti = l; ti < T; ti += LG) {
                        const uint32_t t = t0 + ti;
                        const int32_t cost = (int32_t)nCost[t];
                        const uint32_t sl = nSlot[t];
                        const int16_t *mp = mpair + eb + ti;
                        uint64_t key = ~0ull;
                        for (uint32_t j = 0; j < Pq; j += 4) {
                            const uint32_t j1 = min(j + 1, Pq - 1), j2 = min(j + 2, Pq - 1), j3 = min(j + 3, Pq - 1);
                            const uint2 e0 = bk[p0 + j], e1 = bk[p0 + j1], e2 = bk[p0 + j2], e3 = bk[p0 + j3];
                            const int32_t m0 = mp[j * T], m1 = mp[j1 * T], m2 = mp[j2 * T], m3 = mp[j3 * T];
                            const uint64_t k0 = ((uint64_t)((uint32_t)((int32_t)e0.x + m0) ^ 0x80000000u) << 32) | e0.y;
                            const uint64_t k1 = ((uint64_t)((uint32_t)((int32_t)e1.x + m1) ^ 0x80000000u) << 32) | e1.y;
                            const uint64_t k2 = ((uint64_t)((uint32_t)((int32_t)e2.x + m2) ^ 0x80000000u) << 32) | e2.y;
                            const uint64_t k3 = ((uint64_t)((uint32_t)((int32_t)e3.x + m3) ^ 0x80000000u) << 32) | e3.y;
                            const uint64_t ka = k0 < k1 ? k0 : k1, kb = k2 < k3 ? k2 : k3;
                            const uint64_t kc = ka < kb ? ka : kb;
                            key = kc < key ? kc : key;
                        }
                        const int32_t tot = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u) + cost;
                        const bool ok = Pq != 0 && tot < INF;  // .min(INF) then strict '<' (lattice.rs:135-136)
                        pre[t] = (uint16_t)(ok ? ((uint32_t)key >> 16) : NONE16);
                        if (sl != NONE16) bk[sl].x = (uint32_t)(ok ? tot : INF);
                    }
                }
                __syncthreads();
            }
        }

        // ---- phase 5: backtrace (lattice.rs:144-153), one lane per sentence -----------------------
        if (lane < G) {
            const uint32_t g0 = stab[lane * 4 + 0], cS = stab[lane * 4 + 1];
            uint32_t pos = nb[g0 + cS], pr, K = 0;  // the sentence's EOS node
            while ((pr = pre[pos]) != NONE16 && K <= cS) {
                path[g0 + K++] = (uint16_t)pos;
                if (pr == BOSMARK) break;  // the chain's first node has BOS as predecessor: BOS itself is not emitted
                pos = pr;
            }
            stab[lane * 4 + 3] = (uint16_t)K;
        }
        __syncthreads();
        uint32_t Ksum = 0;
        for (uint32_t s = 0; s < G; ++s) {  // Node -> Token (tokenizer.rs:22-43)
            const uint32_t g0 = stab[s * 4 + 0], cS = stab[s * 4 + 1], bs0 = stab[s * 4 + 2], K = stab[s * 4 + 3];
            const uint64_t sid_ = s0 + s;
            const uint64_t ts = a.offsets[sid_] - a.offsets[0] + sid_;
            const uint32_t Bs = (uint32_t)(a.offsets[sid_ + 1] - a.offsets[sid_]);
            for (uint32_t k = lane; k < K; k += 64) {
                const uint32_t t = path[g0 + K - 1 - k];
                const int32_t sid = nSid[t];
                kgpu_token tk;
                if (sid == 0) {  // Dummy -> "EOS" (tokenizer.rs:27-28,34)
                    tk.id = 0; tk.cls = KGPU_CLASS_DUMMY; tk.position = Bs; tk.start = cS; tk.end = cS + 3; tk.byte_len = 0;
                } else {
                    // a word is never last on the path (EOS is): it ends where its successor starts
                    const uint32_t st = nStart[t], en = nStart[path[g0 + K - 2 - k]], bs = cbyte[st];
                    tk.id = sid > 0 ? sid : -sid;
                    tk.cls = sid > 0 ? KGPU_CLASS_KNOWN : KGPU_CLASS_UNKNOWN;
                    tk.position = bs - bs0; tk.start = st - g0; tk.end = en - g0; tk.byte_len = cbyte[en] - bs;
                }
                a.stage[ts + k] = tk;
            }
            if (lane == 0) { a.status[sid_] = KGPU_SENT_OK; a.tok_count[sid_] = K; }
            Ksum += K;
        }
        if (a.count_work) {
            wT = wave_sum(wT);
            accW[0] += G; accW[1] += Bt; accW[2] += C; accW[3] += wT; accW[4] += N; accW[5] += E; accW[6] += Ksum;
        }
        __syncthreads();
    }
    if (a.count_work && lane == 0)
        for (int k = 0; k < 7; ++k) atomicAdd(&a.ctl->work[k], (unsigned long long)accW[k]);
}

int launch_tokenize_pack(const DictView &d, const BatchArgs &a, const TierIO &io, uint32_t lds_bytes, uint32_t gpack,
                         int n_workgroups, void *stream) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_tokenize_pack, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_tokenize_pack, dim3(n_workgroups), dim3(64), lds_bytes, (hipStream_t)stream, d, a, io, lds_bytes, gpack);
    return (int)hipGetLastError();
}

}  // namespace kgpu

// kanpyo_amd/csrc/kgpu_device.h -- device-side code shared by the tokenize kernels (LDS-resident: kgpu_pool.hip, windowed: kgpu_window.hip,
// general: kgpu_kernels.hip): the trie walks, UTF-8 decoding, and -- once, for both LDS kernels -- the Viterbi sweep step, its DPP minima and the
// connection-cost gather.
#pragma once
#include <hip/hip_runtime.h>

#include "kgpu_internal.h"

namespace kgpu {
namespace dev {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int32_t INF = 1 << 30;               // lattice.rs:117
constexpr uint32_t MAX_UNKNOWN_LEN = 1024;     // lattice.rs:55

__device__ __forceinline__ uint32_t bcast32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t bcast64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// Inclusive prefix sum over the 64 lanes, on the DPP network: four shifted adds inside each row of 16 lanes, then the row
// totals are passed on with row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) -- six VALU ops and no LDS
// crossbar round trips (the ds_bpermute form is six dependent ~100-cycle trips).  Exec must be full.
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {  // lanes whose source is out of range (or whose row is masked) add 0
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t /*lane*/) {
    v = dpp_add<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add<0x118, 0xF>(v);  // row_shr:8
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {  // every lane gets the total
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64), lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {  // every lane gets the total
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63);
}
__device__ __forceinline__ uint32_t ld_l2(const uint32_t *p) {  // bypass the CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Slab {
    uint8_t *ptr;
    uint64_t size;
};

// Grow-only slab owned by this wavefront, carved from the ctx arena.
__device__ __forceinline__ bool slab_ensure(Slab &s, uint64_t need, const BatchArgs &a, uint32_t lane) {
    if (need <= s.size) return true;
    uint64_t want = round_up(need + need / 2 + 256, 256);
    uint64_t off = 0;
    if (lane == 0) off = atomicAdd(&a.ctl->arena_cursor, (unsigned long long)want);
    off = bcast64(off);
    if (off + want > a.arena_bytes) {
        if (lane == 0) atomicExch(&a.ctl->arena_overflow, 1u);
        return false;
    }
    s.ptr = a.arena + off;
    s.size = want;
    return true;
}

// A leaf's payload: (trie id, duplicate count or NONE = "read Morph8::dup of the first record").
__device__ __forceinline__ void leaf_decode(const DictView &d, int32_t base, uint32_t &id, uint32_t &dup) {
    const uint32_t enc = (uint32_t)(-base);
    if (d.leaf_dup) { id = enc & 0x1FFFFFu; dup = enc >> 21; if (dup == 1023u) dup = NONE; }
    else { id = enc; dup = NONE; }
}

// One double-array walk from byte k0 of the sentence (trie/da.rs:155-182).
// F(id, length in chars, duplicate count or NONE) is invoked per match in
// ascending byte length.  Returns nothing; `matched` is set by the callback.
template <class F>
__device__ __forceinline__ uint32_t da_walk(const DictView &d, const uint8_t *text, uint32_t k, uint32_t B,
                                            int32_t base_root, F &&on_match) {
    int32_t p = 1;  // ROOT_ID
    int32_t bp = base_root;
    uint32_t nch = 0, steps = 0;
    for (; k < B; ++k) {
        uint32_t c = text[k];
        ++steps;
        int32_t q = bp + (int32_t)c;
        if ((uint32_t)q >= d.da_len) break;  // negative or past the end: "None" (da.rs:162)
        DaNode nd = d.da[q];
        if (nd.check != p) break;
        p = q;
        bp = nd.base;
        nch += (c & 0xC0) != 0x80;
        int32_t ah = bp;  // + TERMINATOR (da.rs:166)
        if ((uint32_t)ah < d.da_len) {
            DaNode t = d.da[ah];
            if (t.check == p && t.base < 0) { uint32_t id, dup; leaf_decode(d, t.base, id, dup); on_match(id, nch, dup); }
        }
    }
    return steps;
}


// The same walk, started through the first-character table (DictView::first: per BMP code point the node
// reached after its UTF-8 bytes and that node's base, or {0, steps} when the walk dies inside the
// character).  Each iteration then sits on node p at byte k and issues BOTH dependent-free loads
// together: the terminator probe of p (only where a key can end: a character boundary) and the child
// for the next byte -- one memory latency per byte instead of two.  cp = 0xFFFF (not BMP): plain walk.
// kb / kn: byte offsets of this character and of the next one.  Returns the byte steps taken.
template <class F>
__device__ __forceinline__ uint32_t da_walk_first(const DictView &d, const uint8_t *text, uint32_t cp, uint32_t kb, uint32_t kn,
                                                  uint32_t B, int32_t base_root, F &&on_match) {
    if (cp == 0xFFFFu) return da_walk(d, text, kb, B, base_root, on_match);
    const DaNode f = d.first[cp];
    if (f.base == 0) return (uint32_t)f.check;
    int32_t p = f.base, bp = f.check;
    uint32_t k = kn, nstart = 1, steps = kn - kb;
    for (;;) {
        const bool more = k < B;
        const uint32_t c = more ? text[k] : 0u;
        const bool boundary = !more || (c & 0xC0) != 0x80;
        const uint32_t q = (uint32_t)(bp + (int32_t)c);
        const bool doprobe = boundary && (uint32_t)bp < d.da_len;
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

// Two walks of one lane side by side (the long-sentence kernel's count phase: positions i and i + 64): the same steps as
// da_walk_first, but the loads of both walks are issued before either is consumed, so a chunk pair costs the longer of
// its two longest walks instead of their sum.  A walk that is not `on`, or whose character is not in the first table's
// range (cp = 0xFFFF), is left to the caller (plain da_walk).  Returns the byte steps taken by both.
template <class FA, class FB>
__device__ __forceinline__ uint32_t da_walk_first2(const DictView &d, const uint8_t *text, uint32_t B,
                                                   bool onA, uint32_t cpA, uint32_t kbA, uint32_t knA, FA &&matchA,
                                                   bool onB, uint32_t cpB, uint32_t kbB, uint32_t knB, FB &&matchB) {
    struct W { int32_t p, bp; uint32_t k, nstart; bool live; };
    W a{0, 0, knA, 1, false}, b{0, 0, knB, 1, false};
    uint32_t steps = 0;
    DaNode fa{0, 0}, fb{0, 0};
    if (onA) fa = d.first[cpA];
    if (onB) fb = d.first[cpB];
    if (onA) { if (fa.base == 0) steps += (uint32_t)fa.check; else { a.p = fa.base; a.bp = fa.check; a.live = true; steps += knA - kbA; } }
    if (onB) { if (fb.base == 0) steps += (uint32_t)fb.check; else { b.p = fb.base; b.bp = fb.check; b.live = true; steps += knB - kbB; } }
    while (a.live || b.live) {
        // issue: next byte, terminator probe (only where a key can end: a character boundary), child
        const bool moreA = a.live && a.k < B, moreB = b.live && b.k < B;
        const uint32_t cA = moreA ? text[a.k] : 0u, cB = moreB ? text[b.k] : 0u;
        const bool bndA = !moreA || (cA & 0xC0) != 0x80, bndB = !moreB || (cB & 0xC0) != 0x80;
        const uint32_t qA = (uint32_t)(a.bp + (int32_t)cA), qB = (uint32_t)(b.bp + (int32_t)cB);
        const bool prA = a.live && bndA && (uint32_t)a.bp < d.da_len, prB = b.live && bndB && (uint32_t)b.bp < d.da_len;
        const bool nxA = moreA && qA < d.da_len, nxB = moreB && qB < d.da_len;
        DaNode tA{0, 0}, nA{0, 0}, tB{0, 0}, nB{0, 0};
        if (prA) tA = d.da[a.bp];
        if (nxA) nA = d.da[qA];
        if (prB) tB = d.da[b.bp];
        if (nxB) nB = d.da[qB];
        // consume
        if (a.live) {
            if (prA && tA.check == a.p && tA.base < 0) { uint32_t id, dup; leaf_decode(d, tA.base, id, dup); matchA(id, a.nstart, dup); }
            if (!moreA) a.live = false;
            else {
                ++steps;
                if (!nxA || nA.check != a.p) a.live = false;  // da.rs:162-165
                else { a.p = (int32_t)qA; a.bp = nA.base; a.nstart += bndA; ++a.k; }
            }
        }
        if (b.live) {
            if (prB && tB.check == b.p && tB.base < 0) { uint32_t id, dup; leaf_decode(d, tB.base, id, dup); matchB(id, b.nstart, dup); }
            if (!moreB) b.live = false;
            else {
                ++steps;
                if (!nxB || nB.check != b.p) b.live = false;
                else { b.p = (int32_t)qB; b.bp = nB.base; b.nstart += bndB; ++b.k; }
            }
        }
    }
    return steps;
}

// The walk over the character-level array (kgpu_chartrie.cpp): the lane sits on node p (base bp) `depth` characters into the sentence
// from its start position; every round loads the child for the next character's code -- ONE 16-byte load per character tells whether
// the child exists, where its children are and which key ends on it.  code_at(depth): the code of the character `depth` positions after
// the start, 0xFFFF = none (end of the sentence, or a character no key contains).  The load is unconditional: a lane with nothing to
// ask reads slot 0, which has no parent.  (p, bp, leaf) come from CharRec (the root's child for the first character; p == 0: no key
// starts with it), or -- depth = 0, p = 1, bp = the root's base, leaf = 0 -- the walk starts at the root.
template <class CODE, class F>
__device__ __forceinline__ void ct_walk(const DictView &d, int32_t p, int32_t bp, int32_t leaf, CODE &&code_at, F &&on_match, uint32_t depth = 1) {
    if (p == 0) return;
    for (;;) {
        if (leaf < 0) { uint32_t id, dup; leaf_decode(d, leaf, id, dup); on_match(id, depth, dup); }
        const uint32_t c = code_at(depth);
        const uint32_t q = (uint32_t)bp + c;
        const CtNode nx = d.da2[(c != 0xFFFFu && q < d.da2_len) ? q : 0u];
        if (nx.check != p) break;  // da.rs:162-165 (slot 0 has check 0, nodes start at 1)
        p = (int32_t)q;
        bp = nx.base;
        leaf = nx.leaf;
        ++depth;
    }
}
// Two such walks of one lane side by side, both from the root (the long-sentence kernel's count phase: positions i and i + 64): the loads
// of both are issued before either is consumed.
template <class CA, class FA, class CB, class FB>
__device__ __forceinline__ void ct_walk2(const DictView &d, bool onA, CA &&codeA, FA &&matchA, bool onB, CB &&codeB, FB &&matchB) {
    struct W { int32_t p, bp; uint32_t depth; bool live; };
    const int32_t rb = d.da2[1].base;
    W a{1, rb, 0, onA}, b{1, rb, 0, onB};
    while (a.live || b.live) {
        const uint32_t cA = a.live ? codeA(a.depth) : 0xFFFFu, cB = b.live ? codeB(b.depth) : 0xFFFFu;
        const uint32_t qA = (uint32_t)a.bp + cA, qB = (uint32_t)b.bp + cB;
        const CtNode nA = d.da2[(cA != 0xFFFFu && qA < d.da2_len) ? qA : 0u];
        const CtNode nB = d.da2[(cB != 0xFFFFu && qB < d.da2_len) ? qB : 0u];
        if (a.live) {
            if (nA.check != a.p) a.live = false;
            else {
                a.p = (int32_t)qA; a.bp = nA.base; ++a.depth;
                if (nA.leaf < 0) { uint32_t id, dup; leaf_decode(d, nA.leaf, id, dup); matchA(id, a.depth, dup); }
            }
        }
        if (b.live) {
            if (nB.check != b.p) b.live = false;
            else {
                b.p = (int32_t)qB; b.bp = nB.base; ++b.depth;
                if (nB.leaf < 0) { uint32_t id, dup; leaf_decode(d, nB.leaf, id, dup); matchB(id, b.depth, dup); }
            }
        }
    }
}
// Code of a character >= U+FFFF (the BMP table cannot name it): binary search in the dictionary's short list.
__device__ __forceinline__ uint32_t ct_code_nonbmp(const DictView &d, uint32_t cp) {
    uint32_t lo = 0, hi = d.n_nb;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (d.nb_cp[mid] < cp) lo = mid + 1; else hi = mid; }
    return (lo < d.n_nb && d.nb_cp[lo] == cp) ? d.nb_code[lo] : 0xFFFFu;
}
__device__ __forceinline__ uint32_t utf8_cp_at(const uint8_t *t) {  // valid UTF-8 (already checked)
    const uint32_t b = t[0];
    if (b < 0x80) return b;
    if (b < 0xE0) return ((b & 0x1Fu) << 6) | (t[1] & 0x3Fu);
    if (b < 0xF0) return ((b & 0x0Fu) << 12) | ((t[1] & 0x3Fu) << 6) | (t[2] & 0x3Fu);
    return ((b & 0x07u) << 18) | ((t[1] & 0x3Fu) << 12) | ((t[2] & 0x3Fu) << 6) | (t[3] & 0x3Fu);
}

// ---- UTF-8: one lead byte decoded and validated (what Rust's &str guarantees, checked at the C boundary: src/tokenizer.rs:16).  b: the byte at k (a
// lead byte: not 10xxxxxx), B: sentence length, byte_at(k): the sentence's bytes.  Returns the sequence length (1 where the sequence is bad, so that
// the sum of lengths still walks the text) and sets cp / bad.  Overlong forms, surrogates, code points beyond U+10FFFF and truncated sequences are bad.
template <class BY>
__device__ __forceinline__ uint32_t utf8_decode_lead(uint32_t b, uint32_t k, uint32_t B, BY &&byte_at, uint32_t &cp, uint32_t &bad) {
    uint32_t l;
    if (b < 0x80) { l = 1; cp = b; }
    else if (b >= 0xC2 && b <= 0xDF) { l = 2; cp = b & 0x1F; }
    else if ((b & 0xF0) == 0xE0) { l = 3; cp = b & 0x0F; }
    else if (b >= 0xF0 && b <= 0xF4) { l = 4; cp = b & 0x07; }
    else { l = 1; cp = 0; bad = 1; }
    if (k + l > B) { bad = 1; l = 1; }
    for (uint32_t j = 1; j < l; ++j) {
        const uint32_t bb = byte_at(k + j);
        if ((bb & 0xC0) != 0x80) bad = 1;
        cp = (cp << 6) | (bb & 0x3F);
    }
    if (l == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) bad = 1;
    if (l == 4 && (cp < 0x10000 || cp > 0x10FFFF)) bad = 1;
    return l;
}

// ---- LDS access by absolute 32-bit LDS address (the sweep keeps ready-made addresses in its descriptors; going through `array + offset` makes the
// compiler add the array's link-time base -- zero -- to every address, on the VALU)
#define KGPU_LDS(T) __attribute__((address_space(3))) T
// Stride of a target's row in the pair table, in half-words.  KGPU_PAIR_ODD (measurement build, round 5): an odd stride, so that the gather's lane-per-target
// row writes do not all fall into the banks of a power-of-two stride (profiles/experiments/r05_pair_table_stride.txt: what it buys).
#ifdef KGPU_PAIR_ODD
#define KGPU_PSTRIDE(P) ((P) | 1u)
#else
#define KGPU_PSTRIDE(P) (P)
#endif
template <class T> __device__ __forceinline__ T lds_ld(uint32_t addr) { return *(const KGPU_LDS(T) *)(uintptr_t)addr; }
template <class T> __device__ __forceinline__ void lds_st(uint32_t addr, T v) { *(KGPU_LDS(T) *)(uintptr_t)addr = v; }
__device__ __forceinline__ uint2 lds_ld2(uint32_t addr) { const uint64_t v = lds_ld<uint64_t>(addr); return make_uint2((uint32_t)v, (uint32_t)(v >> 32)); }

// Wavefront-level ordering point.  LDS executes one wavefront's instructions in issue order, so data written by one lane is visible to the others at
// the next instruction; this only stops the compiler from moving LDS accesses across it (no s_barrier: the wavefronts of a workgroup are
// independent, and no vmcnt wait: global loads in flight stay in flight).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- group minima on the DPP network over aligned groups of 2^LG lanes (LG <= 4): one DPP-fused v_min per step, every lane gets the result.
// (Exec is full wherever these run -- wave-uniform control flow -- so bound_ctrl never substitutes a zero.)
template <int CTRL>
__device__ __forceinline__ int32_t dpp_i32(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <uint32_t LG>
__device__ __forceinline__ int32_t group_min_i32(int32_t v) {
    if constexpr (LG >= 1) v = min(v, dpp_i32<0xB1>(v));   // quad_perm [1,0,3,2]
    if constexpr (LG >= 2) v = min(v, dpp_i32<0x4E>(v));   // quad_perm [2,3,0,1]
    if constexpr (LG >= 3) v = min(v, dpp_i32<0x141>(v));  // row_half_mirror
    if constexpr (LG >= 4) v = min(v, dpp_i32<0x140>(v));  // row_mirror
    return v;
}
template <uint32_t LG>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {
    if constexpr (LG >= 1) v = min(v, (uint32_t)dpp_i32<0xB1>((int32_t)v));
    if constexpr (LG >= 2) v = min(v, (uint32_t)dpp_i32<0x4E>((int32_t)v));
    if constexpr (LG >= 3) v = min(v, (uint32_t)dpp_i32<0x141>((int32_t)v));
    if constexpr (LG >= 4) v = min(v, (uint32_t)dpp_i32<0x140>((int32_t)v));
    return v;
}
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {  // over the 64 lanes, every lane gets it
    v = group_min_i32<4>(v);
    v = min(v, __shfl_xor(v, 16, 64));
    return min(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = group_min_u32<4>(v);
    v = min(v, (uint32_t)__shfl_xor((int32_t)v, 16, 64));
    return min(v, (uint32_t)__shfl_xor((int32_t)v, 32, 64));
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k) {  // over the 64 lanes, every lane gets it
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t oh = (uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), d, 64), ol = (uint32_t)__shfl_xor((int)(uint32_t)k, d, 64);
        const uint64_t o = ((uint64_t)oh << 32) | ol;
        k = o < k ? o : k;
    }
    return k;
}

// ---- THE Viterbi sweep step of the LDS kernels (kgpu_pool.hip, kgpu_window.hip): one start position, lattice.rs:116-142.
// What a step costs is its dependent chain and its taken branches, not its arithmetic (measured on one wavefront alone: LDS write -> read 86 cycles,
// three dependent DPP minima 43, a taken branch 32, an exec-masked block 64, descriptor read-out + scalar dispatch 52 -- tools/ubench).  So:
//  * per-position descriptors are ready-made LDS byte addresses: D0 = address of nCS[t0] (18 bits) | T (7) << 18 | P (6) << 25 (bit 31 set / D0 == 0:
//    not for this routine), D1 = address of the position's bucket bk[p0] (entries {dp, right | node << 16}), D2 = address of its pair costs (pair
//    (ti, j) at ti * P + j, int16); a_ncs / a_pre / a_bk: the arrays' own addresses (pre[] is parallel to nCS[], half as wide);
//  * three straight-line bodies chosen by ONE scalar compare chain: P <= 8 (pair (ti, j) on lane ti * 8 + j, eight targets per pass), P <= 16 (the same
//    lanes take predecessors j and j + 8), P <= 32 (16 lanes per target, four targets per pass);
//  * every load is unconditional and unclamped (an index past the arrays reads someone else's LDS or zero and is deselected afterwards); an absent
//    candidate is a total no real one reaches (real <= INF + 32767) that still cannot overflow when the word cost is added -- so P = 0 needs no case;
//  * no exec-masked region and no sinks: target groups past T redo target T - 1 -- same loads, same result, same stores to the same addresses;
//  * two DPP group minima: the total, then -- strict '<' over ascending insertion order (lattice.rs:125,136) -- the WHOLE second bucket word among the
//    ties: the node index is its upper half, so the minimum picks the smallest node, and a half-word store writes it;
//  * tot < INF after the add = .min(INF) then strict '<' (lattice.rs:135-136); no fence: one wavefront's DS instructions execute in issue order.
template <uint32_t LG>
__device__ __forceinline__ void sweep_pass2(uint32_t lane, uint32_t tb, uint32_t T, uint32_t P, uint32_t acs, uint32_t apre, uint32_t a_bk, uint32_t D1, uint32_t D2) {
    constexpr uint32_t G = 1u << LG;
    const uint32_t j = lane & (G - 1u), ti = min(tb + (lane >> LG), T - 1u);
    const bool j0v = j < P, j1v = j + G < P;
    const uint32_t cs = lds_ld<uint32_t>(acs + 4 * ti);
    const uint2 e0 = lds_ld2(D1 + 8 * j), e1 = lds_ld2(D1 + 8 * j + 8 * G);
    const uint32_t am = D2 + 2 * (__umul24(ti, KGPU_PSTRIDE(P)) + j);
    const int32_t pc0 = lds_ld<int16_t>(am), pc1 = lds_ld<int16_t>(am + 2 * G);
    __builtin_amdgcn_sched_barrier(0);  // the five reads stay one round trip
    constexpr int32_t ABSENT = 0x7FFEFFFF;
    const int32_t v0 = j0v ? (int32_t)e0.x + pc0 : ABSENT;
    const int32_t v1 = j1v ? (int32_t)e1.x + pc1 : ABSENT;
    const int32_t vmin = group_min_i32<LG>(min(v0, v1));
    const uint32_t n0 = v0 == vmin ? e0.y : 0xFFFFFFFFu, n1 = v1 == vmin ? e1.y : 0xFFFFFFFFu;
    const uint32_t nmin = group_min_u32<LG>(min(n0, n1));
    const int32_t tot = vmin + (int32_t)(int16_t)cs;
    const bool ok = tot < INF;
    lds_st<uint16_t>(apre + 2 * ti, (uint16_t)((ok ? nmin : 0xFFFFFFFFu) >> 16));
    lds_st<uint32_t>(a_bk + 8 * (cs >> 16), (uint32_t)(ok ? tot : INF));
}
__device__ __forceinline__ void sweep_pass1(uint32_t lane, uint32_t tb, uint32_t T, uint32_t P, uint32_t acs, uint32_t apre, uint32_t a_bk, uint32_t D1, uint32_t D2) {
    const uint32_t j = lane & 7u, ti = min(tb + (lane >> 3), T - 1u);   // P <= 8 (87 % of the positions on the cfg 2 corpus): one candidate per lane
    const bool j0v = j < P;
    const uint32_t cs = lds_ld<uint32_t>(acs + 4 * ti);
    const uint2 e0 = lds_ld2(D1 + 8 * j);
    const int32_t pc0 = lds_ld<int16_t>(D2 + 2 * (__umul24(ti, KGPU_PSTRIDE(P)) + j));
    __builtin_amdgcn_sched_barrier(0);
    const int32_t v0 = j0v ? (int32_t)e0.x + pc0 : 0x7FFEFFFF;
    const int32_t vmin = group_min_i32<3>(v0);
    const uint32_t nmin = group_min_u32<3>(v0 == vmin ? e0.y : 0xFFFFFFFFu);
    const int32_t tot = vmin + (int32_t)(int16_t)cs;
    const bool ok = tot < INF;
    lds_st<uint16_t>(apre + 2 * ti, (uint16_t)((ok ? nmin : 0xFFFFFFFFu) >> 16));
    lds_st<uint32_t>(a_bk + 8 * (cs >> 16), (uint32_t)(ok ? tot : INF));
}
// One position with 1 <= T <= 127 targets and P <= 32 predecessors, described by (D0, D1, D2) as above.
__device__ __forceinline__ void sweep_position_fast(uint32_t lane, uint32_t D0, uint32_t D1, uint32_t D2, uint32_t a_ncs, uint32_t a_pre, uint32_t a_bk) {
    const uint32_t acs = D0 & 0x3FFFFu, T = (D0 >> 18) & 127u, P = D0 >> 25;
    const uint32_t apre = a_pre + ((acs - a_ncs) >> 1);  // pre[t0]
    if (P <= 8) {
        sweep_pass1(lane, 0u, T, P, acs, apre, a_bk, D1, D2);
        if (T > 8) for (uint32_t tb = 8; tb < T; tb += 8) sweep_pass1(lane, tb, T, P, acs, apre, a_bk, D1, D2);
    } else if (P <= 16) {
        sweep_pass2<3>(lane, 0u, T, P, acs, apre, a_bk, D1, D2);
        if (T > 8) for (uint32_t tb = 8; tb < T; tb += 8) sweep_pass2<3>(lane, tb, T, P, acs, apre, a_bk, D1, D2);
    } else {
        for (uint32_t tb = 0; tb < T; tb += 4) sweep_pass2<4>(lane, tb, T, P, acs, apre, a_bk, D1, D2);
    }
}
// The three descriptor words of a position for sweep_position_fast (fast = 1 <= T <= 127 and P <= 32 and nothing else in the way; else bit 31).
__device__ __forceinline__ uint32_t sweep_desc0(uint32_t a_ncs, uint32_t t0, uint32_t T, uint32_t P, bool fast) {
    return (a_ncs + 4 * t0) | (fast ? (T << 18) | (P << 25) : 1u << 31);
}

// ---- the connection-cost gather of one target (connection.rs:12-14): M[right(j)][left(t)] for the P predecessors of its position into the LDS pair
// table, eight independent gathers in flight per lane (two groups of four, the second only where the row goes on; the last group of a row padded
// with a repeat of its final entry: ceil(P / 8) dependent rounds).  bk: the position's bucket (its .y carries the right id),
// col: the target's row of the matrix, out: the target's row of the pair table.
__device__ __forceinline__ void gather_target_row(const uint2 *bk, uint32_t P, const int16_t *col, int16_t *out) {
    for (uint32_t j = 0; j < P; j += 8) {
        const uint32_t j1 = min(j + 1, P - 1), j2 = min(j + 2, P - 1), j3 = min(j + 3, P - 1);
        const bool more = j + 4 < P;
        const uint32_t j4 = j + 4, j5 = min(j + 5, P - 1), j6 = min(j + 6, P - 1), j7 = min(j + 7, P - 1);
        const uint32_t r0 = bk[j].y & 0xFFFFu, r1 = bk[j1].y & 0xFFFFu, r2 = bk[j2].y & 0xFFFFu, r3 = bk[j3].y & 0xFFFFu;
        uint32_t r4 = 0, r5 = 0, r6 = 0, r7 = 0;
        if (more) { r4 = bk[j4].y & 0xFFFFu; r5 = bk[j5].y & 0xFFFFu; r6 = bk[j6].y & 0xFFFFu; r7 = bk[j7].y & 0xFFFFu; }
        const int16_t c0 = col[r0], c1 = col[r1], c2 = col[r2], c3 = col[r3];
        int16_t c4 = 0, c5 = 0, c6 = 0, c7 = 0;
        if (more) { c4 = col[r4]; c5 = col[r5]; c6 = col[r6]; c7 = col[r7]; }
        out[j] = c0; out[j1] = c1; out[j2] = c2; out[j3] = c3;
        if (more) { out[j4] = c4; out[j5] = c5; out[j6] = c6; out[j7] = c7; }
    }
}
// The row of the (frequency-ranked) connection matrix a target with left id L reads (connection.rs:12-14).  Layouts that put the pairs of one gather
// instruction into fewer cache lines -- 8 x 8 tiles (rounds 2-3), the transpose -- measure the same as this one (profiles/experiments/r04_matrix_layouts.txt).
__device__ __forceinline__ const int16_t *conn_row(const DictView &d, uint32_t L) { return d.conn + (size_t)d.conn_rows * L; }

// Work-list plumbing shared by the kernels of a launch chain: launch k takes its sentence ids
// from list `in_list` (nullptr = identity over [0, n)) and pushes the ones it does not
// serve (LDS budget, routing) onto the next launch's list.  Work is a
// static grid-stride over the list: the list length is final when the
// kernel starts (same stream), and there is no hot dequeue word -- a single
// contended atomic serialises at ~90 ops/us chip-wide, which is most of a
// 4096-sentence batch.
struct WorkIO {
    const uint32_t *in_list;        // nullptr: sentence id == work index
    const unsigned int *in_count;   // nullptr: a.n
    uint32_t *out_list;             // work list of the next launch (nullptr: none)
    unsigned int *out_count;
    unsigned int *late_count;       // deferrals that happened after the walk (feeds the routing estimate)
};

__device__ __forceinline__ bool work_next(const WorkIO &io, const BatchArgs &a, uint32_t iter, uint64_t &s) {
    const uint64_t i = (uint64_t)blockIdx.x + (uint64_t)iter * gridDim.x;
    if (!io.in_list) { s = i; return i < a.n; }
    const uint64_t n = (uint64_t)bcast32(__hip_atomic_load(io.in_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (i >= n) return false;
    s = (uint64_t)bcast32(io.in_list[i]);
    return true;
}
// Same, for kernels whose workers are wavefronts of larger workgroups: the caller supplies the work index.
__device__ __forceinline__ bool work_next_at(const WorkIO &io, const BatchArgs &a, uint64_t i, uint64_t &s) {
    if (!io.in_list) { s = i; return i < a.n; }
    const uint64_t n = (uint64_t)bcast32(__hip_atomic_load(io.in_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (i >= n) return false;
    s = (uint64_t)bcast32(io.in_list[i]);
    return true;
}
__device__ __forceinline__ void work_defer(const WorkIO &io, uint32_t lane, uint64_t s) {
    if (lane == 0) {
        unsigned int k = atomicAdd(io.out_count, 1u);
        io.out_list[k] = (uint32_t)s;
    }
}

}  // namespace dev
}  // namespace kgpu

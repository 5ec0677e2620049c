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
        CtNode nx = d.da2[(c != 0xFFFFu && q < d.da2_len) ? q : 0u];
        asm volatile("" : "+v"(nx.base), "+v"(nx.check), "+v"(nx.leaf));   // ONE load: left alone, the compiler loads `check` first and the rest behind the test -- a second round trip a character
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
        CtNode nA = d.da2[(cA != 0xFFFFu && qA < d.da2_len) ? qA : 0u];
        CtNode nB = d.da2[(cB != 0xFFFFu && qB < d.da2_len) ? qB : 0u];
        asm volatile("" : "+v"(nA.base), "+v"(nA.check), "+v"(nA.leaf), "+v"(nB.base), "+v"(nB.check), "+v"(nB.leaf));   // (whole records, both in flight: see ct_walk)
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

// ---- STAGE B of the LDS kernels (kgpu_pool.hip, kgpu_window.hip) over a list of TILES: lattice.rs:116-142 with the connection costs of connection.rs:12-14.
// A tile is up to 8 targets x 8 predecessors of one start position, pair (ti, j) on lane 8 ti + j; a position with T targets and P predecessors is
// ceil(T / 8) x ceil(P / 8) tiles -- (a, b) = (target group, predecessor chunk), b fastest: a group's chunks are consecutive, the last one reduces and stores.
// Descriptor (two words, built once per position by the kernel): D0 = LDS address of node[t0 + 8 a] (18 bits) | 8 (Tt - 1) << 18 (6 bits) | 8 (Pt - 1) << 24 (6) |
// first chunk << 30 | last chunk << 31 (the two byte offsets ready-made: one s_bfe each where they are used), D1 = LDS address of bk[p0 + 8 b].  node[t] = {word cost (i16) | BYTE OFFSET of the node's bucket slot << 16 (8 x slot: SLOT_SHIFT; the kernels' LDS budgets keep slots below 8192), byte offset of the node's
// matrix row (left * rows * 2)}; bk[] = bucket entries {dp, 2 * right | node index << 16} (the node index relative to whatever base the kernel uses).
//   GATHER: the lane loads its own connection cost M[right(j)][left(ti)] from the matrix into a REGISTER (byte offset = row offset + 2 * right: one add) -- no
//     pair table in LDS -- and a group of eight tiles is requested while the previous group is swept: the matrix's latency is off the dependency chain.
//   SWEEP: dp of the predecessor + that cost; across a target group's chunks the running lexicographic minimum (total, then the bucket word whose upper half
//     is the node index: strict '<' over ascending insertion order, lattice.rs:125,136); on the last chunk two DPP group minima, the word cost, .min(INF)
//     (tot < INF after the add = .min(INF) then strict '<', lattice.rs:135-136), the stores: dp into the node's bucket slot, the best predecessor into the
//     LOW HALF of node[t].y (the row offset is dead once the node's costs are gathered; 0xFFFF = none).
// No exec mask anywhere: lanes past Tt / Pt repeat the tile's last target / predecessor -- the same loads (the same cache line as their neighbour's), the same
// stores to the same addresses, and a repeated candidate changes no minimum.  (A masked load would have to merge into the register's old value and so wait for
// every load in flight.)  What a step costs is its instruction count and its dependent chain (tools/ubench: LDS write -> read 86 cycles, three dependent DPP
// minima 43, a taken branch 32): no fence per tile -- one wavefront's DS instructions execute in issue order.
struct TileGroup { uint32_t c[8]; };   // per tile of a group: the lane's connection cost (a dword loaded at the cost's 2-byte-aligned address; the low half counts)
constexpr uint32_t TILE_FIRST = 1u << 30, TILE_LAST = 1u << 31;
constexpr uint32_t SLOT_SHIFT = 19, SLOT_MAX = 8191;   // node[t].x = word cost | slot << SLOT_SHIFT: the upper half IS the slot's byte offset in bk[] (one sdwa add in the sweep)
__device__ __forceinline__ uint32_t tile_desc0(uint32_t a_node_t, uint32_t Tt, uint32_t Pt, bool first, bool last) {
    return a_node_t | ((Tt - 1u) << 21) | ((Pt - 1u) << 27) | (first ? TILE_FIRST : 0u) | (last ? TILE_LAST : 0u);
}
// d0 / d1: a window of descriptors, one per lane; the group's tiles are lanes i0 .. i0 + 7.  Issued in the order the sweep consumes them (loads return in
// order: tile u then waits for its own cost only, vmcnt(15 - u) with the next group's eight behind it) and on EVERY path -- a conditional gather makes the
// compiler's count of the loads in flight conservative and a sweep would wait for the next group's loads too.  The kernel's matrix copy is padded by four
// bytes (kgpu_dict_create): a 16-bit load would get its sign extension as a separate instruction behind vmcnt(0) where the value is carried round the loop.
__device__ __forceinline__ void tile_gather8(TileGroup &G, uint32_t d0, uint32_t d1, uint32_t i0, uint32_t tg8, uint32_t j8, const uint8_t *connb) {
    uint32_t lb[8], yy[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t D0 = (uint32_t)__builtin_amdgcn_readlane((int)d0, (int)(i0 + u));
        const uint32_t D1 = (uint32_t)__builtin_amdgcn_readlane((int)d1, (int)(i0 + u));
        lb[u] = lds_ld<uint32_t>((D0 & 0x3FFFFu) + min(tg8, (D0 >> 18) & 0x3Fu) + 4u);
        yy[u] = lds_ld<uint32_t>(D1 + min(j8, (D0 >> 24) & 0x3Fu) + 4u);
    }
    __builtin_amdgcn_sched_barrier(0);  // the sixteen reads are one round trip
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        typedef uint32_t __attribute__((aligned(2))) u32_a2;
        G.c[u] = *(const u32_a2 *)(connb + (lb[u] + (yy[u] & 0xFFFFu)));
        __builtin_amdgcn_sched_barrier(0);
    }
}
// cnt: tiles of the group to sweep (8, or fewer at the end of a list that is not padded: GUARD); rv / ry: the running minimum of the target group in progress.
template <bool GUARD>
__device__ __forceinline__ void tile_sweep8(const TileGroup &G, uint32_t d0, uint32_t d1, uint32_t i0, uint32_t cnt, uint32_t tg8, uint32_t j8, uint32_t a_bk, int32_t &rv, uint32_t &ry) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (GUARD && (uint32_t)u >= cnt) break;
        const uint32_t D0 = (uint32_t)__builtin_amdgcn_readlane((int)d0, (int)(i0 + u));
        const uint32_t na = (D0 & 0x3FFFFu) + min(tg8, (D0 >> 18) & 0x3Fu);
        const uint32_t cs = lds_ld<uint32_t>(na);
        const uint2 e0 = lds_ld2((uint32_t)__builtin_amdgcn_readlane((int)d1, (int)(i0 + u)) + min(j8, (D0 >> 24) & 0x3Fu));
        __builtin_amdgcn_sched_barrier(0);
        const int32_t v0 = (int32_t)e0.x + (int32_t)(int16_t)G.c[u];
        auto reduce_and_store = [&](int32_t bv, uint32_t by) {
            const int32_t vmin = group_min_i32<3>(bv);
            const uint32_t nmin = group_min_u32<3>(bv == vmin ? by : 0xFFFFFFFFu);
            const int32_t tot = vmin + (int32_t)(int16_t)cs;
            const bool ok = tot < INF;
            lds_st<uint16_t>(na + 4u, (uint16_t)((ok ? nmin : 0xFFFFFFFFu) >> 16));
            lds_st<uint32_t>(a_bk + (cs >> 16), (uint32_t)(ok ? tot : INF));
        };
        // (ONE test for "the only chunk of its group" in front of a second copy of the reduction measured 5 % SLOWER: 111-115 against 117-121 M sentences/s)
        {
            if (!(D0 & TILE_FIRST)) {   // a further chunk of the target group: (total, bucket word) lexicographic, one 64-bit compare
                const bool take = (((int64_t)v0 << 32) | e0.y) < (((int64_t)rv << 32) | ry);
                rv = take ? v0 : rv; ry = take ? e0.y : ry;
            } else { rv = v0; ry = e0.y; }
            if (D0 & TILE_LAST) reduce_and_store(rv, ry);
        }
        // (no scheduling barrier between tiles: the compiler keeps the order of LDS accesses it cannot tell apart, and the next tile's address arithmetic and
        // first reads may slide under this tile's second reduction: 118.9 -> 121.6 M sentences/s over three interleaved runs)
    }
}
// Tiles [ta, tb) of the list at `tiles` (LDS), in order.  null_tile: what the lanes past the end of a window hold -- a descriptor whose gather address is
// always valid and that is never swept (the last group's gather reads it for the tiles it does not have: NOT a group already swept, whose nodes' row offsets
// have their best predecessors in the low half by now).  sweep_on = false: measurement (every group's gather, no sweep).
// A window's groups are gathered one ahead of the sweep: while group g is swept out of one register set the costs of group g + 1 arrive in the other, and
// are moved over when the sweep is through (eight register moves a group: by then they have long arrived).  Nothing is gathered behind a window's LAST group
// -- the earlier form alternated the two sets and, to keep the compiler's wait counts exact on every path, gathered "for nothing" behind the end: a third of
// all gathers on a cfg 2 sentence (~64 tiles in windows of 56, as they were: 13 group gathers for 8 groups) -- and only the last group's sweep (code of its own) tests a
// tile against the count, so lists need no padding to whole groups.
__device__ __forceinline__ void tiles_run(const uint2 *tiles, uint32_t ta, uint32_t tb, uint2 null_tile, uint32_t lane, uint32_t a_bk, const uint8_t *connb, bool sweep_on) {
    const uint32_t j8 = 8u * (lane & 7u), tg8 = lane & 0x38u;   // lane = 8 ti + j
    int32_t rv = 0; uint32_t ry = 0;
    for (uint32_t w0 = ta; w0 < tb; w0 += 64) {       // a window of 64 descriptors (eight groups) in registers, read out with v_readlane; lanes past the end: the null tile
        const uint2 dd = w0 + lane < tb ? tiles[w0 + lane] : null_tile;
        const uint32_t d0 = dd.x, d1 = dd.y;
        const uint32_t nt = min(64u, tb - w0), ng = (nt + 7u) >> 3;
        TileGroup GA, GB;
        tile_gather8(GA, d0, d1, 0u, tg8, j8, connb);
        uint32_t g = 0;
        for (; g + 1 < ng; ++g) {
            tile_gather8(GB, d0, d1, 8 * (g + 1), tg8, j8, connb);
            if (sweep_on) tile_sweep8<false>(GA, d0, d1, 8 * g, 8u, tg8, j8, a_bk, rv, ry);
            GA = GB;
        }
        if (sweep_on) tile_sweep8<true>(GA, d0, d1, 8 * g, nt - 8 * g, tg8, j8, a_bk, rv, ry);
        else asm volatile("" :: "v"(GA.c[0]), "v"(GA.c[7]));
    }
}

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

// kanpyo_amd/csrc/kgpu_chartrie.cpp -- the device's character-level copy of the dictionary trie (built once by kgpu_dict_create).
//
// The reference's double array is indexed by BYTES (kanpyo-dict/src/trie/da.rs:155-182): a common-prefix search costs one dependent
// node load per byte, three per Japanese character.  Every walk of the tokenizer starts at a character boundary of valid UTF-8 text
// and keys are UTF-8 strings (Rust `String`s: no key ends inside a character), so the same key set can be walked one CHARACTER at a
// time: this file re-indexes the byte-level trie as a double array over character codes -- one dependent load per character.
//   * codes: 1..n = the distinct characters that occur in keys, ascending by code point;
//   * nodes: {base, check} as in the byte-level array (child of slot p by code c: slot base[p] + c with check == p) plus, in the same
//     16-byte record, the key that ends on the node: leaf = the byte-level terminator child's (re-encoded) base, -(id | dup << 21) or
//     -id (da.rs:118-123, 166) -- the walk learns it from the load that took it there, there is no terminator probe;
//   * slot 0 is never used (a clamped load of it matches nothing), slot 1 is the root.
// The result is an acceleration structure, not a format: it is derived from the caller's double array whatever built that, and the
// byte-level array stays on the device (work counters count the reference's byte steps; dictionaries this file cannot represent --
// 65535 or more distinct characters -- are walked byte by byte as before).
#include <algorithm>
#include <cstdint>
#include <vector>

#include "kgpu_internal.h"

namespace kgpu {

namespace {

struct Edge { uint32_t parent; uint32_t cp; uint32_t child; };   // char-level: parent / child are char-node indices (BFS order)

struct ByteTrie {   // children of every byte-level node, grouped by parent and ascending by byte
    const std::vector<DaNode> &da;
    std::vector<uint32_t> start;    // [len + 1]
    std::vector<uint32_t> slot;     // child slots; the byte is slot - base[parent]
    explicit ByteTrie(const std::vector<DaNode> &d) : da(d) {
        const size_t n = da.size();
        start.assign(n + 1, 0);
        auto edge = [&](size_t q, uint32_t &p) -> bool {
            const int64_t pc = da[q].check;
            if (pc < 1 || (size_t)pc >= n || (size_t)pc == q) return false;
            const int64_t b = da[(size_t)pc].base;
            if (b < 0) return false;  // a leaf has no children
            const int64_t c = (int64_t)q - b;
            if (c < 0 || c > 255) return false;
            p = (uint32_t)pc;
            return true;
        };
        uint32_t p = 0;
        for (size_t q = 0; q < n; ++q) if (edge(q, p)) ++start[p + 1];
        for (size_t i = 0; i < n; ++i) start[i + 1] += start[i];
        slot.resize(start[n]);
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (size_t q = 0; q < n; ++q) if (edge(q, p)) slot[fill[p]++] = (uint32_t)q;  // ascending q = ascending byte within a parent
    }
    template <class F> void children(uint32_t p, uint32_t lo, uint32_t hi, F &&f) const {  // (byte, child slot) with lo <= byte <= hi
        const int64_t b = da[p].base;
        for (uint32_t k = start[p]; k < start[p + 1]; ++k) {
            const uint32_t c = (uint32_t)((int64_t)slot[k] - b);
            if (c >= lo && c <= hi) f(c, slot[k]);
        }
    }
};

}  // namespace

bool build_char_trie(const std::vector<DaNode> &da, const uint8_t *cat, size_t cat_len, CharTrie &out) {
    out = CharTrie{};
    if (da.size() < 2 || da[1].base < 0) return false;
    const ByteTrie bt(da);

    // ---- breadth-first over the characters of the keys: the edges of a char-node are contiguous, ascending by code point ----
    std::vector<uint32_t> byte_of{1u};      // char-node -> byte-level slot (node 0 = the root, slot 1)
    std::vector<int32_t> leaf_of{0};        // char-node -> the terminator child's base (< 0), 0 = no key ends here
    std::vector<uint32_t> first_edge{0u};   // char-node -> its first edge
    std::vector<Edge> edges;
    for (uint32_t cn = 0; cn < byte_of.size(); ++cn) {
        const uint32_t s = byte_of[cn];
        first_edge[cn] = (uint32_t)edges.size();
        auto emit = [&](uint32_t cp, uint32_t q) {
            edges.push_back(Edge{cn, cp, (uint32_t)byte_of.size()});
            byte_of.push_back(q); leaf_of.push_back(0); first_edge.push_back(0);
        };
        bt.children(s, 0, 0, [&](uint32_t, uint32_t q) { if (da[q].base < 0) leaf_of[cn] = da[q].base; });
        bt.children(s, 0x01, 0x7F, [&](uint32_t c, uint32_t q) { emit(c, q); });
        bt.children(s, 0xC2, 0xDF, [&](uint32_t c, uint32_t q) {
            bt.children(q, 0x80, 0xBF, [&](uint32_t c2, uint32_t q2) { emit(((c & 0x1Fu) << 6) | (c2 & 0x3Fu), q2); });
        });
        bt.children(s, 0xE0, 0xEF, [&](uint32_t c, uint32_t q) {
            bt.children(q, 0x80, 0xBF, [&](uint32_t c2, uint32_t q2) {
                bt.children(q2, 0x80, 0xBF, [&](uint32_t c3, uint32_t q3) {
                    const uint32_t cp = ((c & 0x0Fu) << 12) | ((c2 & 0x3Fu) << 6) | (c3 & 0x3Fu);
                    if (cp >= 0x800 && !(cp >= 0xD800 && cp <= 0xDFFF)) emit(cp, q3);  // (what the sentence decoder accepts)
                });
            });
        });
        bt.children(s, 0xF0, 0xF4, [&](uint32_t c, uint32_t q) {
            bt.children(q, 0x80, 0xBF, [&](uint32_t c2, uint32_t q2) {
                bt.children(q2, 0x80, 0xBF, [&](uint32_t c3, uint32_t q3) {
                    bt.children(q3, 0x80, 0xBF, [&](uint32_t c4, uint32_t q4) {
                        const uint32_t cp = ((c & 0x07u) << 18) | ((c2 & 0x3Fu) << 12) | ((c3 & 0x3Fu) << 6) | (c4 & 0x3Fu);
                        if (cp >= 0x10000 && cp <= 0x10FFFF) emit(cp, q4);
                    });
                });
            });
        });
        if (byte_of.size() > (1u << 30)) return false;  // (a cycle in a corrupt array: give up, the byte-level walk copes)
    }
    const uint32_t n_nodes = (uint32_t)byte_of.size();
    first_edge.push_back((uint32_t)edges.size());

    // ---- codes ----
    std::vector<uint32_t> cps;
    cps.reserve(edges.size());
    for (const Edge &e : edges) cps.push_back(e.cp);
    std::sort(cps.begin(), cps.end());
    cps.erase(std::unique(cps.begin(), cps.end()), cps.end());
    if (cps.size() >= 0xFFFFu) return false;
    auto code_of = [&](uint32_t cp) -> uint32_t {
        auto it = std::lower_bound(cps.begin(), cps.end(), cp);
        return (it != cps.end() && *it == cp) ? (uint32_t)(it - cps.begin()) + 1u : 0xFFFFu;
    };
    out.n_codes = (uint32_t)cps.size();

    // ---- placement: first fit over a free-slot chain (next free slot >= q, path-compressed), children of a node at base + code ----
    std::vector<CtNode> &d2 = out.da;
    std::vector<uint32_t> nf;   // nf[q] = q if free, else a later slot to look at
    auto grow = [&](size_t need) {   // nf has one entry more than the array: nf[size] == size, "free, but not there yet"
        if (need <= d2.size()) return;
        const size_t n = std::max<size_t>(need, d2.size() * 2 + 1024), old = d2.size();
        d2.resize(n, CtNode{0, 0, 0, 0});
        nf.resize(n + 1);
        for (size_t q = old + 1; q <= n; ++q) nf[q] = (uint32_t)q;
    };
    nf.assign(1, 0u);
    grow((size_t)out.n_codes + 16);
    auto next_free = [&](uint32_t q) -> uint32_t {
        if (q >= d2.size()) grow((size_t)q + 1);
        uint32_t r = q;
        while (nf[r] != r) r = nf[r];
        for (uint32_t x = q; x != r;) { const uint32_t y = nf[x]; nf[x] = r; x = y; }
        if (r >= d2.size()) grow((size_t)r + 1);
        return r;
    };
    auto take = [&](uint32_t q) { nf[q] = q + 1; };
    take(0); take(1);  // slot 0: never used; slot 1: the root
    std::vector<uint32_t> slot_of(n_nodes, 0);
    slot_of[0] = 1;
    d2[1] = CtNode{0, 0, 0, 0};
    uint32_t cursor = 2;  // multi-child nodes start their search here; it moves on when a region has become too dense to be worth scanning
    std::vector<uint32_t> codes;
    for (uint32_t cn = 0; cn < n_nodes; ++cn) {
        const uint32_t e0 = first_edge[cn], e1 = first_edge[cn + 1];
        codes.clear();
        for (uint32_t e = e0; e < e1; ++e) codes.push_back(code_of(edges[e].cp));
        const uint32_t ps = slot_of[cn];
        d2[ps].leaf = leaf_of[cn];
        if (codes.empty()) continue;  // a key's last character: no children, base stays 0 (slot 0 + code: nothing there has this parent)
        const uint32_t c0 = codes.front(), cmax = codes.back();
        uint32_t q = next_free(std::max(codes.size() > 1 ? cursor : 2u, c0 + 1));
        uint32_t tries = 0;
        for (;;) {
            const uint32_t b = q - c0;
            grow((size_t)b + cmax + 1);
            bool ok = true;
            for (size_t k = 1; k < codes.size() && ok; ++k) { const uint32_t t = b + codes[k]; ok = nf[t] == t; }
            if (ok) {
                d2[ps].base = (int32_t)b;
                size_t k = 0;
                for (uint32_t e = e0; e < e1; ++e, ++k) {
                    const uint32_t t = b + codes[k];
                    d2[t] = CtNode{0, (int32_t)ps, 0, 0};
                    take(t);
                    slot_of[edges[e].child] = t;
                }
                break;
            }
            if (++tries == 64 && codes.size() > 1) { cursor = q; }
            q = next_free(q + 1);
        }
        if (d2.size() > (1ull << 31) - 70000) return false;
    }
    // trim, keeping room for base + any code (the kernels bound-check against the length anyway)
    size_t last = d2.size();
    while (last > 2 && d2[last - 1].check == 0) --last;
    d2.resize(last);

    // ---- per BMP code point: category, code, the root's child ----
    out.rec.assign(65536, CharRec{0, 0, 0xFFFFu, 0, 0, 0});
    const int32_t rb = d2[1].base;
    for (uint32_t cp = 0; cp < 65536; ++cp) {
        CharRec &r = out.rec[cp];
        r.cat = cat_len ? (cp < cat_len ? cat[cp] : cat[0]) : 0;  // char_category_def.rs:33-38
        const uint32_t code = cp == 0xFFFFu ? 0xFFFFu : code_of(cp);  // (U+FFFF shares the kernels' "not in the table" value: it takes the list below)
        r.code = (uint16_t)code;
        if (code != 0xFFFFu) {
            const uint64_t q = (uint64_t)(int64_t)rb + code;
            if (q < d2.size() && d2[(size_t)q].check == 1) { r.base = d2[(size_t)q].base; r.slot = (int32_t)q; r.leaf = d2[(size_t)q].leaf; }
        }
    }
    for (uint32_t cp : cps)
        if (cp >= 0xFFFFu) { out.nb_cp.push_back(cp); out.nb_code.push_back(code_of(cp)); }
    return true;
}

}  // namespace kgpu

// ---- test hook (tests/test_chartrie_cpu.py; not in include/kanpyo_gpu.h): build the character-level array from an index.dict blob's double
// array and run the kernels' walk (ct_walk, kgpu_device.h) on the host for `nq` queries -- checked against the byte-level common-prefix
// search without a GPU.  out: (trie id, length in bytes) pairs, out_offsets[nq + 1]; info: {slots of the array, codes, 1 = built}.
extern "C" int kgpu_debug_chartrie_search(const uint8_t *index_blob, size_t blob_len, const uint8_t *utf8, const uint64_t *offsets, uint64_t nq,
                                          uint32_t *out, uint64_t cap_pairs, uint64_t *out_offsets, uint64_t *info) {
    using namespace kgpu;
    if (!index_blob || blob_len < 8 || !offsets || !out_offsets || !info) return KGPU_ERR_INVALID_ARG;
    uint64_t n = 0;
    for (int k = 0; k < 8; ++k) n |= (uint64_t)index_blob[k] << (8 * k);
    if (n > (blob_len - 8) / 8) return KGPU_ERR_BAD_DICT;
    std::vector<DaNode> da((size_t)n);
    for (size_t i = 0; i < (size_t)n; ++i) {
        uint32_t b = 0, c = 0;
        for (int k = 0; k < 4; ++k) { b |= (uint32_t)index_blob[8 + 8 * i + k] << (8 * k); c |= (uint32_t)index_blob[12 + 8 * i + k] << (8 * k); }
        da[i] = DaNode{(int32_t)b, (int32_t)c};
    }
    CharTrie ct;
    const bool ok = build_char_trie(da, nullptr, 0, ct);
    info[0] = ct.da.size(); info[1] = ct.n_codes; info[2] = ok ? 1 : 0;
    uint64_t np = 0;
    for (uint64_t s = 0; s < nq; ++s) {
        out_offsets[s] = np;
        if (!ok) continue;
        const uint8_t *t = utf8 + offsets[s];
        const uint64_t len = offsets[s + 1] - offsets[s];
        uint64_t k = 0;
        int32_t p = 0, bp = 0;
        auto next_code = [&](uint32_t &code) -> bool {  // the next character's code (0xFFFF: in no key); false: end of text
            if (k >= len) return false;
            const uint8_t b = t[k];
            const uint32_t l = b < 0x80 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
            if (k + l > len) return false;
            uint32_t cp = l == 1 ? b : l == 2 ? (b & 0x1Fu) : l == 3 ? (b & 0x0Fu) : (b & 0x07u);
            for (uint32_t j = 1; j < l; ++j) cp = (cp << 6) | (t[k + j] & 0x3Fu);
            k += l;
            if (cp < 0xFFFFu) code = ct.rec[cp].code;
            else {
                auto it = std::lower_bound(ct.nb_cp.begin(), ct.nb_cp.end(), cp);
                code = (it != ct.nb_cp.end() && *it == cp) ? ct.nb_code[(size_t)(it - ct.nb_cp.begin())] : 0xFFFFu;
            }
            return true;
        };
        uint32_t code = 0;
        if (!next_code(code) || code == 0xFFFFu) continue;
        {   // the root's child (CharRec::slot / base on the device)
            const uint64_t q = (uint64_t)(int64_t)ct.da[1].base + code;
            if (q >= ct.da.size() || ct.da[(size_t)q].check != 1) continue;
            p = (int32_t)q; bp = ct.da[(size_t)q].base;
        }
        for (;;) {  // (ct_walk, kgpu_device.h)
            if (ct.da[(size_t)p].leaf < 0) {
                if (np < cap_pairs) { out[2 * np] = (uint32_t)(-(int64_t)ct.da[(size_t)p].leaf); out[2 * np + 1] = (uint32_t)k; }
                ++np;
            }
            uint32_t c = 0xFFFFu;
            if (!next_code(c) || c == 0xFFFFu) break;
            const uint64_t q = (uint64_t)(uint32_t)bp + c;
            if (q >= ct.da.size() || ct.da[(size_t)q].check != p) break;
            p = (int32_t)q; bp = ct.da[(size_t)q].base;
        }
    }
    out_offsets[nq] = np;
    return np > cap_pairs ? KGPU_ERR_CAPACITY : KGPU_OK;
}

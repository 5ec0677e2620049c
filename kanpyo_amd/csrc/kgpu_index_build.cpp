// kanpyo_amd/csrc/kgpu_index_build.cpp -- host-side IndexTable builder.
//
// Produces the `index.dict` blob (double-array trie + duplicate map) that
// kgpu_dict_create consumes.  The packing has to be byte-identical to the
// reference's so that a dictionary built here and one built by the reference's
// ipa_dict_builder are interchangeable:
//   IndexTable::build      kanpyo-dict/src/index.rs:16-38
//   DoubleArray::add/seek  kanpyo-dict/src/trie/da.rs:43-131
//   truncate / write_dict  kanpyo-dict/src/trie/da.rs:29-35,237-245, index.rs:75-84
// The placement rule is fixed by that contract (first fit from the `left`
// cursor kept in slot 0, cursor advanced when the scanned window is >= 95 %
// occupied, depth-first in byte order).  The implementation here is an
// explicit-stack DFS over key ranges (the sorted keyword list makes every
// trie node a contiguous key range; unsorted-but-prefix-grouped lists such as the
// reference fixture src/tests.rs:9-13 still are), not the reference's recursion over
// Vec<KeywordID>.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kanpyo_gpu.h"
#include "kgpu_internal.h"

namespace {

struct DaNode { int32_t base, check; };

struct Builder {
    std::vector<DaNode> a;
    const uint8_t *keys;
    const std::vector<uint64_t> &koff;  // unique keys
    const std::vector<int64_t> &ids;

    Builder(const uint8_t *k, const std::vector<uint64_t> &o, const std::vector<int64_t> &i)
        : a(50 * 1024, DaNode{0, 0}), keys(k), koff(o), ids(i) {
        a[0].base = 2;  // ROOT_ID + 1 (da.rs:25)
    }
    void grow_to(size_t idx) {  // expand(): double until idx is addressable (da.rs:37-41)
        size_t n = a.size();
        while (idx >= n) n *= 2;
        if (n != a.size()) a.resize(n, DaNode{0, 0});
    }
    // seek (da.rs:43-78): first i >= left with every i+ch slot free.
    size_t seek(const uint8_t *ch, size_t nch) {
        const size_t left = (size_t)a[0].base;
        for (size_t i = left;; ++i) {
            grow_to(i);
            bool ok = true;
            for (size_t k = 0; k < nch; ++k) {
                size_t q = i + ch[k];
                grow_to(q);
                if (a[q].check != 0) { ok = false; break; }
            }
            if (!ok) continue;
            size_t used = 0;
            for (size_t x = left; x <= i; ++x) used += a[x].check != 0;
            if ((double)used / (double)(i - left + 1) >= 0.95) a[0].base = (int32_t)i + 1;
            return i;
        }
    }
    struct Frame { size_t p, depth; uint64_t lo, hi; };
    // Returns false where the reference's `add` would hit its assert (da.rs:106-111):
    // the same byte shows up in two non-adjacent runs of a branch list, i.e. the
    // keyword list is not grouped by prefix (sorted input always is).
    bool build(uint64_t nkeys) {
        std::vector<Frame> st;
        st.push_back({1, 0, 0, nkeys});
        uint8_t ch[257]; uint64_t cs[257], ce[257];
        while (!st.empty()) {
            Frame f = st.back(); st.pop_back();
            grow_to(f.p);
            size_t nch = 0;
            uint64_t seen[4] = {0, 0, 0, 0};
            for (uint64_t k = f.lo; k < f.hi; ++k) {
                size_t len = (size_t)(koff[k + 1] - koff[k]);
                uint8_t c = f.depth < len ? keys[koff[k] + f.depth] : 0;  // TERMINATOR
                if (nch == 0 || ch[nch - 1] != c) {
                    if (seen[c >> 6] & (1ull << (c & 63))) return false;
                    seen[c >> 6] |= 1ull << (c & 63);
                    ch[nch] = c; cs[nch] = k; ++nch;
                }
                ce[nch - 1] = k + 1;
            }
            size_t left = seek(ch, nch);
            a[f.p].base = (int32_t)left;
            for (size_t c = 0; c < nch; ++c) {
                size_t q = left + ch[c];
                a[q].check = (int32_t)f.p;
                if (ch[c] == 0) a[q].base = -(int32_t)ids[f.lo];  // leaf (da.rs:118-123)
            }
            // children depth-first in ascending byte order => push in reverse
            for (size_t c = nch; c-- > 0;) {
                if (ch[c] == 0) continue;
                st.push_back({left + ch[c], f.depth + 1, cs[c], ce[c]});
            }
        }
        return true;
    }
};

}  // namespace

extern "C" int kgpu_index_build(const uint8_t *keys, const uint64_t *key_offsets, uint64_t n,
                                uint8_t **blob, size_t *blob_len) {
    if (!blob || !blob_len || (n && (!keys || !key_offsets))) {
        kgpu::set_error("kgpu_index_build: null argument");
        return KGPU_ERR_INVALID_ARG;
    }
    // IndexTable::build (index.rs:16-38): collapse adjacent duplicates, remember
    // how many extra records share the first id.
    std::vector<uint64_t> koff;  // offsets into `keys` of unique keywords (start), then end
    std::vector<uint64_t> ustart, uend;
    std::vector<int64_t> ids;
    std::vector<std::pair<int64_t, uint64_t>> dup;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t s = key_offsets[i], e = key_offsets[i + 1];
        if (e < s) { kgpu::set_error("kgpu_index_build: offsets not monotone"); return KGPU_ERR_INVALID_ARG; }
        if (!ustart.empty()) {
            uint64_t ps = ustart.back(), pe = uend.back();
            if (pe - ps == e - s && std::memcmp(keys + ps, keys + s, (size_t)(e - s)) == 0) {
                int64_t first = ids.back();
                if (!dup.empty() && dup.back().first == first) dup.back().second++;
                else dup.push_back({first, 1});
                continue;
            }
        }
        ustart.push_back(s); uend.push_back(e); ids.push_back((int64_t)i + 1);
    }
    // pack unique keys contiguously so a key range is [koff[k], koff[k+1])
    std::string packed;
    koff.reserve(ustart.size() + 1);
    for (size_t k = 0; k < ustart.size(); ++k) {
        koff.push_back(packed.size());
        packed.append((const char *)keys + ustart[k], (size_t)(uend[k] - ustart[k]));
    }
    koff.push_back(packed.size());

    Builder b((const uint8_t *)packed.data(), koff, ids);
    if (!b.build(ustart.size())) {
        kgpu::set_error("kgpu_index_build: keywords are not grouped by prefix (sort them; the reference panics here, da.rs:106)");
        return KGPU_ERR_INVALID_ARG;
    }
    size_t len = b.a.size();  // truncate (da.rs:29-35)
    while (len > 1 && b.a[len - 1].check == 0) --len;

    size_t bytes = 8 + len * 8 + 8 + dup.size() * 16;
    uint8_t *out = (uint8_t *)std::malloc(bytes);
    if (!out) { kgpu::set_error("kgpu_index_build: out of memory"); return KGPU_ERR_INTERNAL; }
    size_t o = 0;
    uint64_t u = len; std::memcpy(out + o, &u, 8); o += 8;
    std::memcpy(out + o, b.a.data(), len * 8); o += len * 8;
    u = dup.size(); std::memcpy(out + o, &u, 8); o += 8;
    for (auto &kv : dup) {
        std::memcpy(out + o, &kv.first, 8); std::memcpy(out + o + 8, &kv.second, 8); o += 16;
    }
    *blob = out; *blob_len = bytes;
    return KGPU_OK;
}

extern "C" void kgpu_free(void *p) { std::free(p); }

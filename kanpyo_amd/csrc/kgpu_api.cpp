// kanpyo_amd/csrc/kgpu_api.cpp -- host runtime behind include/kanpyo_gpu.h.
//
// Owns: blob parsing + validation (the panics of the reference's hot path are
// turned into KGPU_ERR_BAD_DICT at create time), the one-time dictionary upload
// to HBM, per-ctx stream / scratch arena / staging, the launch sequence
// (tokenize -> scan -> compact) and the host-buffer convenience entry point.
// There is NO CPU fallback: without a HIP device every entry point that would
// compute returns KGPU_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>
#include <dirent.h>
#include <pthread.h>
#include <sched.h>
#include <linux/futex.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <climits>
#include <memory>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "kgpu_runtime.h"

namespace kgpu {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

}  // namespace kgpu

using namespace kgpu;

namespace {

struct Reader {
    const uint8_t *p; size_t n, at = 0; bool bad = false;
    Reader(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    template <class T> T get() {
        T v{};
        if (at + sizeof(T) > n) { bad = true; return v; }
        std::memcpy(&v, p + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    size_t left() const { return n - at; }
};


}  // namespace

// Test-only environment hooks.  getenv is not thread-safe against setenv, and kgpu_tokenize_batch may be called from many
// threads: the hooks are read under a mutex, ONCE per process -- unless KGPU_TEST_HOOKS_REREAD is set (tests/conftest.py sets
// it: the tests flip the hooks between calls).
static bool env_flag_now(const char *name) { const char *e = getenv(name); return e && *e && *e != '0'; }
TestHooks kgpu::test_hooks() {
    static std::mutex mu;
    static TestHooks cur;
    static bool init = false;
    std::lock_guard<std::mutex> g(mu);
    if (!init || env_flag_now("KGPU_TEST_HOOKS_REREAD")) {
        init = true;
        cur = TestHooks{};
        cur.no_small_calls = env_flag_now("KGPU_NO_SMALL_CALLS");
        cur.legacy_host_path = env_flag_now("KGPU_HOST_LEGACY");
        cur.plain_leaves = env_flag_now("KGPU_PLAIN_LEAVES");
        cur.byte_trie = env_flag_now("KGPU_BYTE_TRIE");  // no character-level copy of the trie: every kernel walks the bytes
        if (const char *e = getenv("KGPU_HOST_DEPTH")) cur.depth = strtoull(e, nullptr, 10);
        if (const char *e = getenv("KGPU_HOST_CHUNK_BYTES")) cur.chunk_bytes = strtoull(e, nullptr, 10);
        if (const char *e = getenv("KGPU_HOST_CHUNK_SENTS")) cur.chunk_sents = strtoull(e, nullptr, 10);
        if (const char *e = getenv("KGPU_MULTI_CHUNK_SENTS")) cur.multi_chunk_sents = strtoull(e, nullptr, 10);  // kgpu_tokenize_batch_multi: sentences per device and chunk (tests: many small super-chunks)
    }
    return cur;
}

// Every slot of a double array has ONE parent (its check), so the child edges can only loop back through the root: a root that is itself some
// node's child means a corrupt array whose breadth-first re-indexing (kgpu_chartrie.cpp) would never end.  Such a dictionary is walked byte by byte
// (a walk is bounded by the sentence, whatever the array looks like).
static bool char_trie_buildable(const std::vector<DaNode> &da) {
    if (da.size() < 2) return false;
    const int64_t p = da[1].check;
    if (p < 1 || (size_t)p >= da.size() || p == 1) return true;
    const int64_t b = da[(size_t)p].base, c = 1 - b;
    return !(b >= 0 && c >= 0 && c <= 255);
}
// Test hook (tests/test_chartrie_cpu.py): would kgpu_dict_create build the character-level copy for this index.dict blob?
extern "C" int kgpu_debug_char_trie_usable(const uint8_t *index_blob, size_t blob_len) {
    if (!index_blob || blob_len < 8) return -1;
    uint64_t n = 0;
    std::memcpy(&n, index_blob, 8);
    if (n > (blob_len - 8) / 8) return -1;
    std::vector<DaNode> da((size_t)n);
    if (n) std::memcpy(da.data(), index_blob + 8, (size_t)n * 8);
    CharTrie ct;
    return char_trie_buildable(da) && build_char_trie(da, nullptr, 0, ct) ? 1 : 0;
}

static Combiner *combiner_new();
static void combiner_delete(Combiner *c);

static void dict_release(kgpu_dict *d) {
    if (d->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    combiner_delete(d->combiner);
    (void)hipSetDevice(d->device);
    for (hipStream_t st : d->streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipStream_t st : d->long_streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (void *p : d->allocs) (void)hipFree(p);
    delete d;
}

extern "C" const char *kgpu_last_error(void) { return g_err; }

// Concurrent launches need hardware queues: HIP gives a process GPU_MAX_HW_QUEUES of them (default 4, of which its streams get three), and
// four launches side by side are the optimum of this library.  The variable is read when the HIP runtime initialises, so the
// library sets it itself when it is loaded early enough -- before any HIP / HSA call of the process, i.e. while /dev/kfd is not open yet --
// and the caller has not set it.  Loaded too late (or with the variable set below 5) it runs on three streams and says so
// (kgpu_plan_info.streams, and a warning in kgpu_last_error after kgpu_dict_create).
static bool g_queues_ok = false;
static int g_queues = 4;   // hardware queues the HIP runtime of this process has (or will have): HIP's default unless the variable says otherwise
#ifndef KGPU_HW_QUEUES
#define KGPU_HW_QUEUES 16
#define KGPU_HW_QUEUES_STR "16"
#endif
static bool kfd_is_open() {   // has this process opened the compute driver already (= has a HIP / HSA runtime been initialised)?
    DIR *dir = opendir("/proc/self/fd");
    if (!dir) return true;     // cannot tell: assume the worst (three streams) rather than count on queues that may not be there
    bool found = false;
    char link[300], target[256];
    while (const dirent *e = readdir(dir)) {
        if (e->d_name[0] == '.') continue;
        snprintf(link, sizeof link, "/proc/self/fd/%s", e->d_name);
        const ssize_t k = readlink(link, target, sizeof target - 1);
        if (k > 0) { target[k] = 0; if (strcmp(target, "/dev/kfd") == 0) { found = true; break; } }
    }
    closedir(dir);
    return found;
}
__attribute__((constructor)) static void kgpu_preinit() {
    if (const char *off = getenv("KGPU_NO_PREINIT")) if (*off && *off != '0') return;   // the host does not want its environment touched at load time: three streams unless it sets the variable itself
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    if (e) { g_queues_ok = atoi(e) >= 5; g_queues = std::max(1, atoi(e)); return; }   // the caller's choice stands
    if (kfd_is_open()) return;                          // the runtime is up already: too late, three streams
    setenv("GPU_MAX_HW_QUEUES", KGPU_HW_QUEUES_STR, 0);
    g_queues_ok = true;
    g_queues = KGPU_HW_QUEUES;
}
static unsigned planned_streams() {
    static const unsigned n = getenv("KGPU_STREAMS") && atoi(getenv("KGPU_STREAMS")) > 0 ? (unsigned)atoi(getenv("KGPU_STREAMS")) : (g_queues_ok ? 4u : 3u);
    return n;
}
// Streams for chains that start with the windowed kernel, beside the shared ones: what the hardware queues leave (16 queues: eight; 8: two; the default 4: none --
// such chains then stay on the shared streams).  KGPU_LONG_STREAMS overrides (0 = off).
static unsigned planned_long_streams() {
    static const unsigned n = [] {
        if (const char *e = getenv("KGPU_LONG_STREAMS")) return (unsigned)std::max(0, std::min(16, atoi(e)));
        const int spare = g_queues - 6;
        return (unsigned)std::max(0, std::min(8, spare));
    }();
    return n;
}

extern "C" int kgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------- dictionary

template <class T>
static int upload(kgpu_dict *d, const std::vector<T> &h, const T **out) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
    HIPCHECK(hipMalloc(&p, bytes));
    d->allocs.push_back(p);
    if (!h.empty()) HIPCHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    d->info.device_bytes += bytes;
    *out = (const T *)p;
    return KGPU_OK;
}

static int parse_morphs(Reader &r, std::vector<Morph8> &out, const char *what) {
    int64_t n = r.get<int64_t>();  // morph.rs:74-78
    if (r.bad || n < 0 || (uint64_t)n > r.left() / 6) { set_error("%s: truncated morph block", what); return KGPU_ERR_BAD_DICT; }
    out.resize((size_t)n);
    for (auto &m : out) { m.left = r.get<int16_t>(); m.right = r.get<int16_t>(); m.cost = r.get<int16_t>(); m.dup = 0; }
    return KGPU_OK;
}

extern "C" int kgpu_dict_create(const kgpu_dict_blobs *b, int device, kgpu_dict **out) {
    if (!b || !out) { set_error("kgpu_dict_create: null argument"); return KGPU_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!b->index_dict || !b->connection_dict || !b->morph_dict || !b->unk_dict || !b->char_category ||
        (!b->invoke_list && b->invoke_len) || (!b->group_list && b->group_len)) {
        set_error("kgpu_dict_create: null blob");
        return KGPU_ERR_INVALID_ARG;
    }
    // ---- parse (layouts: SURVEY.md App. B) ----
    std::vector<DaNode> da;
    std::vector<std::pair<int64_t, uint64_t>> dup;
    {
        Reader r(b->index_dict, b->index_len);  // trie/da.rs:220-236, index.rs:57-73
        uint64_t n = r.get<uint64_t>();
        if (r.bad || n > r.left() / 8 || n >= (1ull << 31)) { set_error("index.dict: bad double-array length"); return KGPU_ERR_BAD_DICT; }
        da.resize((size_t)n);
        if (n) { std::memcpy(da.data(), r.p + r.at, (size_t)n * 8); r.at += (size_t)n * 8; }
        uint64_t m = r.get<uint64_t>();
        if (r.bad || m > r.left() / 16) { set_error("index.dict: bad duplicate map"); return KGPU_ERR_BAD_DICT; }
        dup.resize((size_t)m);
        for (auto &kv : dup) { kv.first = r.get<int64_t>(); kv.second = r.get<uint64_t>(); }
    }
    // DoubleArray::search_common_prefix_of indexes self.0[1] unconditionally
    // (da.rs:156,161) and panics on a 1-element array (empty keyword list);
    // here such a trie simply matches nothing.
    while (da.size() < 2) da.push_back(DaNode{0, 0});

    uint64_t rows, cols;
    std::vector<int16_t> conn;
    {
        Reader r(b->connection_dict, b->connection_len);  // connection.rs:28-42
        rows = r.get<uint64_t>(); cols = r.get<uint64_t>();
        if (r.bad || rows >= (1ull << 31) || cols >= (1ull << 31) || (rows && cols > r.left() / 2 / rows)) {
            set_error("connection.dict: truncated"); return KGPU_ERR_BAD_DICT;
        }
        if (rows * cols >= (1ull << 32)) { set_error("connection.dict: matrix too large"); return KGPU_ERR_BAD_DICT; }
        conn.resize((size_t)(rows * cols));
        if (!conn.empty()) std::memcpy(conn.data(), r.p + r.at, conn.size() * 2);
    }
    std::vector<Morph8> morphs, unk_morphs;
    {
        Reader r(b->morph_dict, b->morph_len);
        int rc = parse_morphs(r, morphs, "morph.dict");
        if (rc) return rc;
    }
    std::vector<CatInfo> cinfo(256, CatInfo{0, 0, 0, 0});
    {
        Reader r(b->unk_dict, b->unk_len);  // unk_dict.rs:75-99
        uint64_t k = r.get<uint64_t>();
        if (r.bad || k > r.left() / 17) { set_error("unk.dict: truncated"); return KGPU_ERR_BAD_DICT; }
        struct E { uint8_t cat; int64_t first; uint64_t count; };
        std::vector<E> ents((size_t)k);
        for (auto &e : ents) { e.cat = r.get<uint8_t>(); e.first = r.get<int64_t>(); e.count = r.get<uint64_t>(); }
        int rc = parse_morphs(r, unk_morphs, "unk.dict");
        if (rc) return rc;
        for (auto &e : ents) {
            // lattice.rs:195 unk_dict.morphs[id - 1] must be in bounds for every id the entry yields
            if (e.count && (e.first < 1 || (uint64_t)e.first - 1 + e.count > unk_morphs.size())) {
                set_error("unk.dict: category %u maps to morph ids %lld..+%llu outside 1..%zu (reference would panic, lattice.rs:195)",
                          e.cat, (long long)e.first, (unsigned long long)e.count, unk_morphs.size());
                return KGPU_ERR_BAD_DICT;
            }
            cinfo[e.cat].flags |= CAT_HAS_UNK;
            cinfo[e.cat].unk_first = (int32_t)e.first;
            cinfo[e.cat].unk_count = (uint32_t)e.count;
        }
    }
    if (b->char_category_len == 0) { set_error("char_category: empty table (reference would panic, char_category_def.rs:37)"); return KGPU_ERR_BAD_DICT; }
    for (size_t i = 0; i < b->invoke_len && i < 256; ++i) if (b->invoke_list[i]) cinfo[i].flags |= CAT_INVOKE;
    for (size_t i = 0; i < b->group_len && i < 256; ++i) if (b->group_list[i]) cinfo[i].flags |= CAT_GROUP;
    {
        bool seen[256] = {false};
        for (size_t i = 0; i < b->char_category_len; ++i) seen[b->char_category[i]] = true;
        for (int c = 0; c < 256; ++c)
            if (seen[c] && (size_t)c >= b->invoke_len) {
                set_error("char_category: category %d has no invoke_list entry (reference would panic, lattice.rs:54)", c);
                return KGPU_ERR_BAD_DICT;
            }
    }
    // duplicate counts ride in the first record's padding
    for (auto &kv : dup) {
        if (kv.first < 1 || (uint64_t)kv.first > morphs.size() || kv.second > 65535 ||
            (uint64_t)kv.first + kv.second > morphs.size()) {
            set_error("index.dict: duplicate entry (%lld,+%llu) outside 1..%zu morphs", (long long)kv.first,
                      (unsigned long long)kv.second, morphs.size());
            return KGPU_ERR_BAD_DICT;
        }
        morphs[(size_t)kv.first - 1].dup = (uint16_t)kv.second;
    }
    // every leaf id (and its duplicates) must index morphs (lattice.rs:182)
    for (size_t a = 0; a < da.size(); ++a) {
        const DaNode &nd = da[a];
        if (nd.base < 0 && nd.check > 0 && (size_t)nd.check < da.size() && da[(size_t)nd.check].base == (int32_t)a) {
            int64_t id = -(int64_t)nd.base;
            if (id > (int64_t)morphs.size()) {
                set_error("index.dict: keyword id %lld has no morph (reference would panic, lattice.rs:182)", (long long)id);
                return KGPU_ERR_BAD_DICT;
            }
        }
    }
    // ConnectionTable::get(right, left) = data[rows*left + right] (connection.rs:12-14):
    // every (right, left) combination of the dictionary must stay in bounds.
    {
        int64_t max_l = 0, max_r = 0;
        auto scan = [&](const std::vector<Morph8> &v) {
            for (auto &m : v) {
                if (m.left < 0 || m.right < 0) return false;
                max_l = std::max<int64_t>(max_l, m.left); max_r = std::max<int64_t>(max_r, m.right);
            }
            return true;
        };
        if (!scan(morphs) || !scan(unk_morphs)) { set_error("morph: negative context id (reference would panic, connection.rs:13)"); return KGPU_ERR_BAD_DICT; }
        if ((uint64_t)max_l * rows + (uint64_t)max_r >= conn.size()) {
            set_error("connection.dict: %llux%llu matrix does not cover left_id %lld / right_id %lld (reference would panic, connection.rs:13)",
                      (unsigned long long)rows, (unsigned long long)cols, (long long)max_l, (long long)max_r);
            return KGPU_ERR_BAD_DICT;
        }
    }

    // ---- frequency-rank the context ids -------------------------------------------------
    // Context ids are only ever used to index the connection matrix (they are not part of a
    // Token), so they can be renumbered freely.  Ranking both id spaces by how many dictionary
    // records carry them puts the ids the lattice meets most often at the low indices: for a
    // target (one matrix row of 2.6 KB) all its frequent predecessors then sit in the row's first
    // cache line, and the frequent rows' first lines stay L1-resident.  The sweep's dominant L2
    // consumer is exactly this gather (connection.rs:12-14 once per relaxation).
    uint32_t bos_right = 0, eos_left = 0;
    std::vector<uint32_t> rank_r, rank_l;  // id -> rank, kept (inverted) for kgpu_lattice_dump
    {
        bool in_range = true;  // remap only when every id is a plain (row, col) index
        for (auto *v : {&morphs, &unk_morphs})
            for (auto &m : *v) if ((uint64_t)m.right >= rows || (uint64_t)m.left >= cols) in_range = false;
        if (in_range && rows && cols && rows < 65536 && cols < 65536) {
            std::vector<uint64_t> fr(rows, 0), fl(cols, 0);
            for (auto *v : {&morphs, &unk_morphs}) for (auto &m : *v) { fr[m.right]++; fl[m.left]++; }
            fr[0] += morphs.size(); fl[0] += morphs.size();  // BOS/EOS take part in every sentence
            auto rank = [](const std::vector<uint64_t> &f) {
                std::vector<uint32_t> order(f.size()), map(f.size());
                for (uint32_t i = 0; i < f.size(); ++i) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return f[x] > f[y]; });
                for (uint32_t k = 0; k < order.size(); ++k) map[order[k]] = k;
                return map;
            };
            const std::vector<uint32_t> rmap = rank(fr), lmap = rank(fl);
            std::vector<int16_t> c2(conn.size());
            for (uint64_t l = 0; l < cols; ++l)
                for (uint64_t r = 0; r < rows; ++r) c2[(size_t)(lmap[l] * rows + rmap[r])] = conn[(size_t)(l * rows + r)];
            conn.swap(c2);
            for (auto *v : {&morphs, &unk_morphs})
                for (auto &m : *v) { m.right = (int16_t)rmap[m.right]; m.left = (int16_t)lmap[m.left]; }
            bos_right = rmap[0]; eos_left = lmap[0];
            rank_r = rmap; rank_l = lmap;
        }
    }

    // ---- the device copy of the double array carries the duplicate counts in its leaves ----
    // A leaf (reached through the terminator byte, trie/da.rs:118-123) stores base = -id.  The walk has to load that node
    // anyway, and the record count of the surface (index.rs:46-51) is the next thing it needs: with ids below 2^21 the spare
    // bits hold it (1023 = larger, look it up), and one dependent load per match disappears from the walk.
    uint32_t leaf_dup = 0;
    if (morphs.size() < (1u << 21) && !test_hooks().plain_leaves /* tests: the layout of a dictionary with 2^21 morphs or more */) {
        leaf_dup = 1;
        for (size_t a2 = 0; a2 < da.size(); ++a2) {
            DaNode &nd = da[a2];
            if (nd.base < 0 && nd.check > 0 && (size_t)nd.check < da.size() && da[(size_t)nd.check].base == (int32_t)a2) {
                const uint32_t id = (uint32_t)(-(int64_t)nd.base);
                const uint32_t dupc = std::min<uint32_t>(morphs[id - 1].dup, 1023u);
                nd.base = -(int32_t)(id | (dupc << 21));
            }
        }
    }

    // ---- upload once to HBM ----
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("kgpu_dict_create: no HIP device (the HIP path has no CPU fallback)");
        return KGPU_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("kgpu_dict_create: device %d out of range (0..%d)", device, ndev - 1); return KGPU_ERR_INVALID_ARG; }
    HIPCHECK(hipSetDevice(device));
    kgpu_dict *d = new kgpu_dict();
    d->combiner = combiner_new();
    d->device = device;
    d->right_of_rank.resize(rank_r.size()); d->left_of_rank.resize(rank_l.size());
    for (uint32_t i = 0; i < rank_r.size(); ++i) d->right_of_rank[rank_r[i]] = i;
    for (uint32_t i = 0; i < rank_l.size(); ++i) d->left_of_rank[rank_l[i]] = i;
    std::vector<uint8_t> cat(b->char_category, b->char_category + b->char_category_len);
    int rc;
    // First-character jump table: the walk from every start position begins with a whole
    // character, so the 1-3 dependent node loads of its UTF-8 bytes (trie/da.rs:159-165) are
    // memoised per BMP code point.  Keys are UTF-8 strings, so no key ends inside a character.
    std::vector<DaNode> first(65536);
    for (uint32_t cp = 0; cp < 65536; ++cp) {
        uint8_t b[3]; int len;
        if (cp < 0x80) { b[0] = (uint8_t)cp; len = 1; }
        else if (cp < 0x800) { b[0] = 0xC0 | (cp >> 6); b[1] = 0x80 | (cp & 0x3F); len = 2; }
        else { b[0] = 0xE0 | (cp >> 12); b[1] = 0x80 | ((cp >> 6) & 0x3F); b[2] = 0x80 | (cp & 0x3F); len = 3; }
        int32_t pp = 1, steps = 0;
        bool ok = true;
        for (int k = 0; k < len; ++k) {
            ++steps;
            const int64_t q = (int64_t)da[(size_t)pp].base + b[k];
            if (q < 0 || q >= (int64_t)da.size() || da[(size_t)q].check != pp) { ok = false; break; }
            pp = (int32_t)q;
        }
        first[cp] = ok ? DaNode{pp, da[(size_t)pp].base} : DaNode{0, steps};
    }
    // Character-level copy of the trie (kgpu_chartrie.cpp): one dependent load per character instead of one per byte.
    CharTrie ct;
    const bool have_ct = !test_hooks().byte_trie && char_trie_buildable(da) && build_char_trie(da, cat.data(), cat.size(), ct);
    if (have_ct) {
        if ((rc = upload(d, ct.da, &d->view.da2)) || (rc = upload(d, ct.rec, &d->view.crec)) ||
            (rc = upload(d, ct.nb_cp, &d->view.nb_cp)) || (rc = upload(d, ct.nb_code, &d->view.nb_code))) {
            kgpu_dict_destroy(d);
            return rc;
        }
        d->view.da2_len = (uint32_t)ct.da.size();
        d->view.n_nb = (uint32_t)ct.nb_cp.size();
    }
    // ONE table for the known and the unknown words' records (DictView: unk_morph == morph + n_morph): a node's record is one index away whichever kind it is
    std::vector<Morph8> all_morphs(morphs);
    all_morphs.insert(all_morphs.end(), unk_morphs.begin(), unk_morphs.end());
    conn.resize(conn.size() + 2, 0);   // the pool kernel reads a cost with a dword load at its (2-byte-aligned) address: the last element's load stays inside the allocation
    if ((rc = upload(d, first, &d->view.first)) ||
        (rc = upload(d, da, &d->view.da)) || (rc = upload(d, all_morphs, &d->view.morph)) || (rc = upload(d, conn, &d->view.conn)) ||
        (rc = upload(d, cat, &d->view.cat)) || (rc = upload(d, cinfo, &d->view.cinfo))) {
        kgpu_dict_destroy(d);
        return rc;
    }
    d->view.da_len = (uint32_t)da.size();
    d->view.leaf_dup = leaf_dup;
    d->view.n_morph = (uint32_t)morphs.size();
    d->view.n_unk_morph = (uint32_t)unk_morphs.size();
    d->view.unk_morph = d->view.morph + morphs.size();
    d->view.conn_rows = (uint32_t)rows;
    d->view.bos_right = bos_right; d->view.eos_left = eos_left;
    d->view.cat_len = (uint32_t)std::min<size_t>(cat.size(), 0x110000);
    d->info.da_len = da.size(); d->info.n_morphs = morphs.size(); d->info.n_unk_morphs = unk_morphs.size();
    d->info.conn_rows = rows; d->info.conn_cols = cols; d->info.device = device;
    *out = d;
    g_err[0] = 0;
    if (planned_streams() < 4)   // not an error: the handle is good, the message is there for whoever looks
        set_error("warning: running on %u streams -- GPU_MAX_HW_QUEUES was %s when the HIP runtime initialised; set GPU_MAX_HW_QUEUES=16 in the environment "
                  "(or load this library before the first HIP call) for the full rate of concurrent batches", planned_streams(),
                  getenv("GPU_MAX_HW_QUEUES") ? "below 5" : "unset");
    return KGPU_OK;
}

extern "C" void kgpu_dict_destroy(kgpu_dict *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    std::vector<kgpu_ctx *> pooled;
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        pooled.swap(d->pool);
    }
    for (auto *c : pooled) kgpu_ctx_destroy(c);
    dict_release(d);
}

extern "C" int kgpu_dict_get_info(const kgpu_dict *d, kgpu_dict_info *out) {
    if (!d || !out) { set_error("kgpu_dict_get_info: null argument"); return KGPU_ERR_INVALID_ARG; }
    *out = d->info;
    return KGPU_OK;
}

// ----------------------------------------------------------------------- ctx

static constexpr size_t ARENA_INITIAL = 1ull << 28;  // 256 MiB; only the general (HBM-scratch) kernel uses it, grows x2 on demand
static constexpr size_t ARENA_MAX = 1ull << 37;      // 128 GiB

extern "C" int kgpu_ctx_create(kgpu_dict *d, void *hip_stream, kgpu_ctx **out) {
    if (!d || !out) { set_error("kgpu_ctx_create: null argument"); return KGPU_ERR_INVALID_ARG; }
    *out = nullptr;
    HIPCHECK(hipSetDevice(d->device));
    kgpu_ctx *c = new kgpu_ctx();
    c->dict = d;
    d->refs.fetch_add(1, std::memory_order_relaxed);
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = true; }
    else {
        std::lock_guard<std::mutex> g(d->pool_mu);
        // Streams that really run side by side: HIP gives a process GPU_MAX_HW_QUEUES hardware queues (default 4), of
        // which its streams get one fewer; a stream beyond that shares a queue and unbalances them (4 streams on the
        // default: 52 M sentences/s instead of 68).  Four concurrent launches are the optimum (71.5; five: 58), so:
        // 4 streams when the process was started with GPU_MAX_HW_QUEUES >= 5, else 3.  KGPU_STREAMS overrides.
        const unsigned n_streams = planned_streams();
        if (d->streams.size() < n_streams) {
            hipStream_t st = nullptr;
            hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); kgpu_ctx_destroy(c); return KGPU_ERR_HIP; }
            d->streams.push_back(st);
            c->stream = st;
        } else {
            c->stream = d->streams[d->next_stream++ % d->streams.size()];
        }
        c->short_stream = c->stream;
    }
    if (hipMalloc((void **)&c->d_ctl, sizeof(Control)) != hipSuccess ||
        hipHostMalloc((void **)&c->h_ctl, sizeof(Control), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->h_ctl_dev, c->h_ctl, 0) != hipSuccess) {
        set_error("kgpu_ctx_create: control block allocation failed");
        kgpu_ctx_destroy(c);
        return KGPU_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->switch_ev, hipEventDisableTiming) != hipSuccess) {
        set_error("kgpu_ctx_create: hipEventCreate failed");
        kgpu_ctx_destroy(c);
        return KGPU_ERR_HIP;
    }
    c->plan = default_launch_plan(d->device);
    *out = c;
    return KGPU_OK;
}

extern "C" void kgpu_ctx_destroy(kgpu_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->dict->device);
    if (c->pending && c->done_ev) (void)hipEventSynchronize(c->done_ev);
    if (c->counted_long) { c->dict->long_sentences_in_flight.fetch_sub(c->counted_long, std::memory_order_relaxed); c->counted_long = 0; }
    if (c->done_ev) (void)hipEventDestroy(c->done_ev);
    if (c->switch_ev) (void)hipEventDestroy(c->switch_ev);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    c->arena.release(); c->ovf.release(); c->stat_slots.release(); c->stage.release(); c->tok_count.release();
    c->in_utf8.release(); c->in_off.release(); c->out_tok.release(); c->out_off.release(); c->out_status.release();
    c->pin_in.release(); c->pin_out.release(); c->in_block.release();
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    if (c->sm_host) (void)hipHostFree(c->sm_host);
    kgpu_dict *d = c->dict;
    delete c;
    dict_release(d);
}

static int next_event(kgpu_ctx *c, hipEvent_t *ev) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    *ev = c->ev_pool[c->ev_used++];
    return KGPU_OK;
}

// Which chain the next batch gets, and on which stream.  A batch of long sentences (by its average length: the host knows n and the bytes, not the
// lengths) starts with the windowed kernel -- the pool launch in front of it would only route: a thousand 40 KB workgroups that each look at four sentences
// and pass them on, waiting for LDS on a chip full of single-wavefront workgroups (cfg 5, 8 in flight: 2.97 -> 3.40 Gchar/s without it) -- and runs on a stream
// of the long set, one per context, so that eight such launches overlap instead of four (-> 3.96; both: profiles/experiments/r05_long_chains.txt).
// The context's previous batch is complete here (kgpu_ctx_sync), so switching streams needs no ordering for the context's own buffers; whatever the
// host-buffer paths queued on the old stream for THIS batch (their H2D copy) is ordered in front by an event.
static int ctx_pick_chain(kgpu_ctx *c, uint64_t n, uint64_t total_bytes, bool dump) {
    const unsigned lim = c->plan.window_first_bytes;   // (KGPU_WINDOW_FIRST, read with the launch plan when the context is created)
    c->window_first = lim && n && c->plan.n_pools && c->plan.window_lds_bytes && !dump && c->stop_after == 0 && !c->no_window && total_bytes >= (uint64_t)lim * n;
    // Dense lattices: when four reservations of the learnt size (LDS bytes per input byte, steered by the redo rate: kgpu_ctx_sync) do not fit the pool, the
    // batch's pool workgroups get THREE wavefronts -- a fourth sentence would only wait for pages (the dense-lattice dictionary, natural density N/C = 8.6:
    // 57.8 -> 62.2 M sentences/s; cfg 2's reservations fit and it stays at four: three would cost it 21 %; profiles/experiments/r06_tile_sweep.txt)
    {
        static const unsigned pct = [] { const char *e = getenv("KGPU_ROOMY_PCT"); const int v = e ? atoi(e) : 92; return (unsigned)(v < 0 ? 0 : v); }();   // (measurement; 0 = never)
        const uint64_t est1 = n ? ((total_bytes / n) * c->dict->est_q8.load(std::memory_order_relaxed) >> 8) + 768u : 0u;
        c->roomy = pct && n && c->plan.n_pools && c->plan.pool_limit_auto && c->plan.pool_waves[0] == 4 && 4u * est1 * 100u > (uint64_t)c->plan.pool_bytes[0] * pct;
    }
    if (c->own_stream) { c->h2d_queued = false; return KGPU_OK; }
    hipStream_t want = c->short_stream;
    // ... and so does a pool-first chain whose last batch sent an eighth or more of its sentences on to the windowed kernel: its launches behind the pool
    // kernel are the long ones (cfg 3 in batches of 4096: 15.4 -> 17.5 M sentences/s on eight streams; a pool-ONLY chain loses there: cfg 2 101 -> 86)
    if ((c->window_first || c->long_share) && planned_long_streams()) {
        if (!c->long_stream) {
            kgpu_dict *d = c->dict;
            std::lock_guard<std::mutex> g(d->pool_mu);
            if (d->long_streams.size() < planned_long_streams()) {
                hipStream_t st = nullptr;
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) d->long_streams.push_back(st);
                else (void)hipGetLastError();
            }
            if (!d->long_streams.empty()) c->long_stream = d->long_streams[d->next_long++ % d->long_streams.size()];
        }
        if (c->long_stream) want = c->long_stream;
    }
    if (want != c->stream) {
        // only what THIS call queued on the old stream (a host-buffer path's H2D copy) has to be in front of the batch; without it no ordering is needed (the
        // context's previous batch is complete) -- and an event on a shared stream would put the batch behind the other contexts' whole backlog there
        if (c->h2d_queued) {
            HIPCHECK(hipEventRecord(c->switch_ev, c->stream));
            HIPCHECK(hipStreamWaitEvent(want, c->switch_ev, 0));
        }
        c->stream = want;
    }
    c->h2d_queued = false;
    return KGPU_OK;
}

static int enqueue(kgpu_ctx *c, const BatchArgs &a) {
    // The Control block is zero here: the previous launch's scan kernel left it so.
    if (c->ctl_dirty) HIPCHECK(hipMemsetAsync(c->d_ctl, 0, sizeof(Control), c->stream));
    c->ctl_dirty = true;  // until this enqueue is through
    hipEvent_t e0 = nullptr, ef = nullptr, e1 = nullptr, e2 = nullptr;
    int rc;
    const bool timed = c->profiling && (c->launch_seq++ % c->event_every) == 0;
    if (timed) {
        if ((rc = next_event(c, &e0)) || (rc = next_event(c, &ef)) || (rc = next_event(c, &e1)) || (rc = next_event(c, &e2))) return rc;
        HIPCHECK(hipEventRecord(e0, c->stream));
    }
    if (a.n) {
        const int pools_now = (c->window_first && !c->no_window) ? 0 : c->dict->big_pool_batches.load(std::memory_order_relaxed) > 0 ? c->plan.n_pools : std::min(c->plan.n_pools, 1);
        c->last_pools = pools_now;
        // The windowed kernel is in the chain while recent batches left the pools sentences (starts armed) -- an empty launch of a few thousand
        // workgroups behind a chip full of long-running wavefronts is not free -- or always, without a pool kernel in front of it.
        const bool window_now = c->plan.window_lds_bytes && !c->no_window && !a.dump_lattice && c->stop_after == 0 &&
                                (pools_now == 0 || c->dict->window_batches.load(std::memory_order_relaxed) > 0);
        c->last_window = window_now;
        // The general kernel closes the chain when nothing else is in it, in ablation / dump runs, and while recent batches left it sentences;
        // otherwise nothing does -- a sentence that needed more shows in the last work list's count, and kgpu_ctx_sync launches what is missing
        // over that list.
        c->last_tail = (pools_now == 0 && !window_now) || c->stop_after != 0 || a.dump_lattice || c->no_window ||
                       c->dict->tail_batches.load(std::memory_order_relaxed) > 0;
        // Two wavefronts per sentence (the windowed kernel's team form) when the list is short against the chip: the sentences of this batch AND of the
        // window-first batches in flight lately are at most twice the form's resident workgroups -- a lone batch of 1000 documents fills a quarter of the
        // single-wavefront slots and each document is one wavefront's chain; with four or more such batches in flight the ordinary form is the better use of the LDS.
        const int team_mode = c->plan.window_team_mode;   // KGPU_WINDOW_TEAM: 0 never, 2 whenever possible, default by the load
        bool team_now = false;
        if (pools_now == 0 && window_now && c->plan.window_team_workgroups > 0 && team_mode != 0) {
            if (!c->counted_long) { c->counted_long = (int)std::min<uint64_t>(a.n, 1u << 30); c->dict->long_sentences_in_flight.fetch_add(c->counted_long, std::memory_order_relaxed); }
            const int cur = c->dict->long_sentences_in_flight.load(std::memory_order_relaxed), old = c->dict->long_peak.load(std::memory_order_relaxed);
            const int peak = std::max(cur, old - old / 8);
            c->dict->long_peak.store(peak, std::memory_order_relaxed);
            // measured on cfg 5 (1000 documents per batch, Mchar/s, ordinary / team form): 1 in flight 1084 / 1495, 2: 1957 / 2153, 4: 3376 / 2372, 8: 4145 / 2405
            team_now = team_mode == 2 || peak <= 2 * c->plan.window_team_workgroups;
        }
        c->last_team = team_now;
        // The pool's SHAPE for this batch.  A pool-only chain keeps four wavefronts on 40 KB (cfg 2 100.9 M sentences/s; two on 20 KB: 98.4-99.3, the dense dictionary
        // 53.7 -> 50.8).  A chain that holds a windowed launch shares the chip with thousands of 10 KB single-wavefront workgroups that run for a millisecond: a
        // workgroup of two wavefronts on 20 KB finds its LDS and its wavefront slots far sooner than one of four on 40 KB -- cfg 3 at 4096 per batch 18.5 -> 21.9 M
        // sentences/s, at 65 536 23.9 -> 25.3 -- and in small batches (one sentence per wavefront slot: the pool launch lasts as long as its longest sentence) it
        // routes a little earlier (56 of its 64 pages of 312 B instead of all).  profiles/experiments/r05_long_chains.txt, sections 5 and 8.
        LaunchPlan pl = c->plan;
        if (pools_now > 0 && c->long_share && pl.pool_limit_auto && pl.alt_pool_workgroups > 0) {
            pl.pool_bytes[0] = pl.alt_pool_bytes; pl.pool_waves[0] = pl.alt_pool_waves; pl.pool_workgroups[0] = pl.alt_pool_workgroups;
            pl.pool_max_pages[0] = a.n <= 4u * 4096u ? 56u : 64u;
        }
        else if (pools_now > 0 && c->roomy) pl.pool_waves[0] = 3;   // (the same pools, the same grid: a workgroup's tickets hand its share out to three wavefronts)
        // The windowed launch behind the pools: as many workgroups as the last batch's share of routed sentences suggests (+ a quarter), not the chip's 4096 -- the
        // list is strided, so an estimate that is too small only makes a workgroup take a second sentence (the context's first batch gets the full grid).
        static const int grid_mode = [] { const char *e = getenv("KGPU_WINDOW_GRID"); return e ? atoi(e) : -1; }();   // (measurement: 0 = always the full grid)
        int window_grid = 0;
        if (grid_mode != 0 && pools_now > 0 && window_now && c->rt.batches > 0)
            window_grid = (int)std::min<uint64_t>(1u << 20, std::max<uint64_t>(256, ((a.n * c->win_share_q8) >> 8) * 5 / 4 + 64));
        hipError_t e = (hipError_t)launch_tokenize(c->dict->view, a, pl, pools_now, c->stop_after, c->stream, ef, window_now, c->last_tail, team_now, window_grid);
        if (e != hipSuccess) { set_error("k_tokenize launch: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    } else if (timed) HIPCHECK(hipEventRecord(ef, c->stream));
    // (the two small kernels behind a pool-only chain on a partner stream of each shared stream, so that the shared stream goes on with the next pool launch at once:
    // measured with 16 hardware queues, cfg 2 100.5 -> 72.9 M sentences/s -- whatever lets a fifth pool launch start early loses, profiles/experiments/r05_long_chains.txt)
    if (timed) HIPCHECK(hipEventRecord(e1, c->stream));
    {
        c->h_ctl->pack_overflow = 0;  // set by the compaction's workgroups in the host copy directly; this context's previous batch has been synced
        static const int scan_small = [] { const char *e = getenv("KGPU_SCAN_SMALL"); return e ? atoi(e) : -1; }();   // (measurement: 0 never, 1 always; default: behind chains with a windowed launch)
        const bool small_wgs = scan_small >= 0 ? scan_small == 1 : (a.n && c->last_window && (c->last_pools == 0 || c->long_share));
        hipError_t e = (hipError_t)launch_scan_compact(a, c->h_ctl_dev, c->stream, small_wgs, scan_small == 2);
        if (e != hipSuccess) { set_error("scan/compact launch: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    }
    if (timed) HIPCHECK(hipEventRecord(e2, c->stream));
    HIPCHECK(hipEventRecord(c->done_ev, c->stream));
    c->ctl_dirty = false;
    c->last = a;
    c->pending = true;
    return KGPU_OK;
}

int kgpu::tokenize_device_impl(kgpu_ctx *c, const uint8_t *d_utf8, const uint64_t *d_offsets, uint64_t n, uint64_t total_bytes,
                                kgpu_token *d_tokens, kgpu_token8 *d_tokens8, uint32_t *d_first, uint8_t *status8, uint64_t *toff8, uint64_t token_capacity,
                                uint64_t *d_tok_offsets, uint8_t *d_status, const char *who) {
    if (!c || !d_offsets || !d_tok_offsets || (n && !d_status) || (total_bytes && !d_utf8) ||
        (token_capacity && !d_tokens && !d_tokens8) || (d_tokens8 && n && !d_first)) {
        set_error("%s: null argument", who);
        return KGPU_ERR_INVALID_ARG;
    }
    if (n >= (1ull << 32) - 1) { set_error("%s: more than 2^32-2 sentences in one batch; split it", who); return KGPU_ERR_INVALID_ARG; }
    if (total_bytes >= (1ull << 32)) { set_error("%s: batch larger than 4 GiB; split it", who); return KGPU_ERR_INVALID_ARG; }
    HIPCHECK(hipSetDevice(c->dict->device));
    int rc;
    if (c->pending && (rc = kgpu_ctx_sync(c, nullptr)) != KGPU_OK && rc != KGPU_ERR_CAPACITY) return rc;
    if ((rc = ctx_pick_chain(c, n, total_bytes, false))) return rc;
    if ((rc = c->arena.ensure(ARENA_INITIAL)) || (rc = c->stage.ensure((size_t)(total_bytes + n + 1) * sizeof(kgpu_token) + 64)) ||
        (rc = c->tok_count.ensure((size_t)(n + 1) * 4)) ||
        (rc = c->ovf.ensure((size_t)(n + 1) * 4 * 4)))
        return rc;
    BatchArgs a{};
    a.utf8 = d_utf8; a.offsets = d_offsets; a.n = n; a.ctl = c->d_ctl;
    a.arena = (uint8_t *)c->arena.p; a.arena_bytes = c->arena.bytes;
    a.stage = (kgpu_token *)c->stage.p;
    a.tok_count = (uint32_t *)c->tok_count.p;
    a.status = d_status; a.out = d_tokens; a.out_cap = token_capacity; a.tok_offsets = d_tok_offsets;
    a.out8 = d_tokens8; a.first8 = d_first; a.status8 = status8; a.toff8 = toff8;
    a.count_work = c->count_work ? (c->count_no_t ? 3u : 1u) : 0u;
#ifdef KGPU_STEP_TIMING
    const bool want_stats = true;
#else
    const bool want_stats = c->count_work;
#endif
    if (want_stats) {
        if (!c->stat_slots.p) {
            if ((rc = c->stat_slots.ensure((size_t)STAT_SLOTS * STAT_WORDS * 8))) return rc;
            HIPCHECK(hipMemsetAsync(c->stat_slots.p, 0, (size_t)STAT_SLOTS * STAT_WORDS * 8, c->stream));
        }
        a.stat_slots = (unsigned long long *)c->stat_slots.p;
    }
    a.est_q8 = c->dict->est_q8.load(std::memory_order_relaxed);
    for (int k = 0; k < 4; ++k) a.ovf[k] = (uint32_t *)c->ovf.p + (size_t)k * (n + 1);
    return enqueue(c, a);
}

// The long-sentence kernel alone over work list `li` of the pending batch (which the chain left unserved), then scan + compaction again.
static int enqueue_tail(kgpu_ctx *c, int li) {
    const BatchArgs &a = c->last;
    // (the scan kernel zeroed the control block after publishing it; the completed batch is behind us on the stream)
    HIPCHECK(hipMemcpyAsync(&c->d_ctl->ovf_count[li], &c->tail_count, sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    c->ctl_dirty = true;
    const bool window_was_in = c->last_window;
    hipError_t e = (hipError_t)launch_tail_only(c->dict->view, a, c->plan, li, window_was_in || c->no_window, c->stream);
    if (e != hipSuccess) { set_error("tail launch: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    if (!window_was_in && !c->no_window && c->plan.window_lds_bytes) c->last_window = true;
    c->last_tail = true;
    c->h_ctl->pack_overflow = 0;
    e = (hipError_t)launch_scan_compact(a, c->h_ctl_dev, c->stream);
    if (e != hipSuccess) { set_error("scan/compact launch: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    HIPCHECK(hipEventRecord(c->done_ev, c->stream));
    c->ctl_dirty = false;
    c->pending = true;
    return KGPU_OK;
}

extern "C" int kgpu_tokenize_device(kgpu_ctx *c, const uint8_t *d_utf8, const uint64_t *d_offsets, uint64_t n,
                                    uint64_t total_bytes, kgpu_token *d_tokens, uint64_t token_capacity,
                                    uint64_t *d_tok_offsets, uint8_t *d_status) {
    return tokenize_device_impl(c, d_utf8, d_offsets, n, total_bytes, d_tokens, nullptr, nullptr, nullptr, nullptr, token_capacity, d_tok_offsets, d_status,
                                "kgpu_tokenize_device");
}

extern "C" int kgpu_tokenize_device_compact(kgpu_ctx *c, const uint8_t *d_utf8, const uint64_t *d_offsets, uint64_t n,
                                            uint64_t total_bytes, kgpu_token8 *d_tokens8, uint64_t token_capacity,
                                            uint32_t *d_first, uint64_t *d_tok_offsets, uint8_t *d_status) {
    if (token_capacity && !d_tokens8) { set_error("kgpu_tokenize_device_compact: null argument"); return KGPU_ERR_INVALID_ARG; }
    return tokenize_device_impl(c, d_utf8, d_offsets, n, total_bytes, nullptr, d_tokens8, d_first, nullptr, nullptr, token_capacity, d_tok_offsets, d_status,
                                "kgpu_tokenize_device_compact");
}

// Host side of the 8-byte records: position / start are running sums over the sentence (include/kanpyo_gpu.h, kgpu_token8).
extern "C" void kgpu_expand_tokens(const kgpu_token8 *in, const uint64_t *tok_offsets, const uint32_t *first, uint64_t n, kgpu_token *out) {
    const bool stream = n && expand_stream_wanted(tok_offsets[n] - tok_offsets[0]);
    expand_tokens(in, tok_offsets, first, n, out, stream);
    if (stream) expand_fence();
}

// The pending batch is over (completed or given up): the context is free, its share of the dictionary's long sentences in flight is returned.
static void ctx_retire(kgpu_ctx *c) {
    c->pending = false;
    if (c->counted_long) { c->dict->long_sentences_in_flight.fetch_sub(c->counted_long, std::memory_order_relaxed); c->counted_long = 0; }
}

extern "C" int kgpu_ctx_sync(kgpu_ctx *c, uint64_t *n_tokens) {
    if (!c) { set_error("kgpu_ctx_sync: null ctx"); return KGPU_ERR_INVALID_ARG; }
    HIPCHECK(hipSetDevice(c->dict->device));
    for (;;) {
        if (!c->pending) { if (n_tokens) *n_tokens = 0; return KGPU_OK; }
        HIPCHECK(hipEventSynchronize(c->done_ev));  // this context's batch only: later work on a shared stream is not waited for
        if (c->h_ctl->window_fail && !c->h_ctl->arena_overflow && !c->no_window) {
            // the windowed kernel met a sentence it cannot hold and had no list to hand it on to: the batch once more without it
            c->no_window = true;
            c->rt.window_reruns++;
            static const bool wtrace = env_flag_now("KGPU_WINDOW_TRACE");
            if (wtrace) {
                const unsigned long long *w = c->h_ctl->phase;
                fprintf(stderr, "window kernel handed back sentences of a batch of %llu: seeds out of range %llu, prefix overflow %llu, node chunks %llu, FIFO full %llu, "
                                "FIFO chunks %llu, FIFO order %llu, carry list %llu, window LDS %llu\n", (unsigned long long)c->last.n, w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
            }
            // the rerun counts everything again: drop what this run left in the per-wavefront slots
            if (c->last.stat_slots) HIPCHECK(hipMemsetAsync(c->last.stat_slots, 0, (size_t)STAT_SLOTS * STAT_WORDS * 8, c->stream));
            c->tail_pass = false;
            int rc = enqueue(c, c->last);
            c->no_window = false;
            if (rc) { ctx_retire(c); return rc; }
            continue;
        }
        {   // KGPU_WINDOW_TRACE=1: why the windowed kernel handed sentences on (Control::phase[1..9] count its reasons in non-profiling runs)
            static const bool wtrace = env_flag_now("KGPU_WINDOW_TRACE");
            if (wtrace && c->last_window && !c->last.count_work && c->h_ctl->ovf_count[c->last_pools] > 0) {
                const unsigned long long *w = c->h_ctl->phase;
                fprintf(stderr, "windowed kernel: %u of %u sentences handed on (batch of %llu): seeds out of range %llu, prefix overflow %llu, node chunks %llu, FIFO full %llu, "
                                "FIFO chunks %llu, FIFO order %llu, carry list %llu, window LDS %llu\n", c->h_ctl->ovf_count[c->last_pools], c->last_pools ? c->h_ctl->ovf_count[c->last_pools - 1] : (unsigned)c->last.n,
                        (unsigned long long)c->last.n, w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
            }
        }
        const int li_last = c->last_pools - 1 + (c->last_window ? 1 : 0) + (c->last_team ? 1 : 0);   // the list the chain ended on
        if (!c->last_tail && c->last.n && li_last >= 0 && !c->h_ctl->arena_overflow && c->h_ctl->ovf_count[li_last] > 0) {
            // The chain ended without its tail and a sentence needed it: ONLY what is missing (the windowed kernel if it was not in the chain,
            // then the general kernel), over the last work list (still in device memory; its length goes back into the control block the scan
            // kernel zeroed), then scan + compaction once more.  The pool kernel's work is not repeated: a corpus with a sparse but steady
            // share of long sentences pays a small launch per such batch, not the batch twice.  What the first pass counted (routing,
            // estimate feedback, work counters) is kept and merged below.
            c->rt.tail_reruns++;
            c->tail_saved = *c->h_ctl;
            c->tail_pass = true;
            c->tail_li = li_last;
            c->tail_had_window = c->last_window;
            c->tail_count = c->h_ctl->ovf_count[li_last];
            int rc = enqueue_tail(c, li_last);
            if (rc) { ctx_retire(c); c->tail_pass = false; return rc; }
            continue;
        }
        if (c->h_ctl->arena_overflow) {
            // a lattice did not fit the scratch arena: grow it and redo the batch
            size_t want = c->arena.bytes * 2;
            if (want > ARENA_MAX) { set_error("scratch arena exceeded %zu bytes", ARENA_MAX); ctx_retire(c); return KGPU_ERR_INTERNAL; }
            int rc = c->arena.ensure(want);
            if (rc) { ctx_retire(c); return rc; }
            BatchArgs a = c->last;
            a.arena = (uint8_t *)c->arena.p; a.arena_bytes = c->arena.bytes;
            c->rt.arena_regrows++;
            c->tail_pass = false;  // (the whole batch runs again: nothing of an earlier pass is merged)
            // the rerun counts everything again: drop what the aborted run left in the per-wavefront slots (ctl->work went with the control block)
            if (a.stat_slots) HIPCHECK(hipMemsetAsync(a.stat_slots, 0, (size_t)STAT_SLOTS * STAT_WORDS * 8, c->stream));
            if ((rc = enqueue(c, a))) { ctx_retire(c); return rc; }
            continue;
        }
        break;
    }
    ctx_retire(c);
    bool first_window = c->last_window, first_tail = c->last_tail;   // what the FIRST pass of this batch had in its chain (the arming below decays on that)
    if (c->tail_pass) {   // the published block is the tail pass's: put back what the first pass had counted
        c->tail_pass = false;
        first_window = c->tail_had_window; first_tail = false;
        Control &h = *c->h_ctl;
        const Control &sv = c->tail_saved;
        for (int k = 0; k < 4; ++k) { if (k <= c->tail_li) h.ovf_count[k] = sv.ovf_count[k]; h.late_count[k] += sv.late_count[k]; }   // lists up to the one the tail served are the first pass's
        for (int k = 0; k < 7; ++k) h.work[k] += sv.work[k];
        for (int k = 0; k < 10; ++k) h.phase[k] += sv.phase[k];
        h.pack_overflow |= sv.pack_overflow;
    }
    c->rt.batches++; c->rt.sentences += c->last.n;
    for (int k = 0; k < 4; ++k) { c->rt.deferred[k] += c->h_ctl->ovf_count[k]; c->rt.redone[k] += c->h_ctl->late_count[k]; }
    if (c->last_window && c->last.n) c->rt.long_launches++;
    if (c->last.n && c->last_pools > 0) {
        // arming of the launches behind the pools: what the pools left arms the windowed kernel, what the windowed kernel left arms the general
        // kernel; eight clean batches disarm (a wrong guess costs one small extra launch over the batch's last list, not the batch)
        const unsigned pool_left = c->h_ctl->ovf_count[c->last_pools - 1];
        if (c->plan.window_lds_bytes) {
            if (pool_left > 0) c->dict->window_batches.store(64, std::memory_order_relaxed);
            else if (first_window) c->dict->window_batches.fetch_sub(8, std::memory_order_relaxed);
        }
        if (c->last_window || !c->plan.window_lds_bytes) {   // (a batch whose pools left sentences while the windowed kernel was disarmed re-arms that one, not this)
            const unsigned behind = c->last_window ? c->h_ctl->ovf_count[c->last_pools] : pool_left;   // what the last launch in front of the general kernel left
            if (behind > 0) c->dict->tail_batches.store(64, std::memory_order_relaxed);
            else if (first_tail) c->dict->tail_batches.fetch_sub(8, std::memory_order_relaxed);
        }
    }
    if (c->last.n && c->last_pools > 0) {
        c->win_share_q8 = c->plan.window_lds_bytes ? (uint32_t)std::min<uint64_t>(256, (uint64_t)c->h_ctl->ovf_count[c->last_pools - 1] * 256 / c->last.n) : 0u;
        c->long_share = c->long_share ? c->win_share_q8 >= 16 : c->win_share_q8 >= 32;   // (entered at an eighth, left below a sixteenth: a share that hovers around the limit does not flap between streams)
    }
    if (c->last.n && c->last_pools == 0 && c->last_window && c->plan.n_pools) {
        // a chain that started with the windowed kernel: what it left arms the general kernel behind it, as above
        if (c->h_ctl->ovf_count[c->last_team ? 1 : 0] > 0) c->dict->tail_batches.store(64, std::memory_order_relaxed);
        else if (first_tail) c->dict->tail_batches.fetch_sub(8, std::memory_order_relaxed);
    }
    if (c->last.n && c->plan.n_pools && c->last_pools > 0) {
        // The pool kernel reserves est LDS bytes per input byte up front: a reservation that proves
        // too small costs a redo (late_count), one that is too large only idles pages until the
        // lattice is known -- steer for a redo rate of 1-3 %.  Applied to the value the batch ran with; races between
        // contexts only lose an adjustment.
        if (c->plan.n_pools > 1) {
            if (c->h_ctl->ovf_count[0] > 0) c->dict->big_pool_batches.store(64, std::memory_order_relaxed);
            else if (c->last_pools > 1) c->dict->big_pool_batches.fetch_sub(1, std::memory_order_relaxed);
        }
        const unsigned late = c->h_ctl->late_count[0];
        uint32_t est = c->last.est_q8;
        if ((uint64_t)late * 4 > c->last.n) est += est / 4;
        else if ((uint64_t)late * 32 > c->last.n) est += est / 16;
        else if ((uint64_t)late * 100 < c->last.n) est -= est / 128;
        est = std::min<uint32_t>(std::max<uint32_t>(est, 16 * 256), 1024 * 256);
        if (est != c->last.est_q8) c->dict->est_q8.store(est, std::memory_order_relaxed);
    }
    if (c->profiling) {
        for (size_t i = 0; i + 4 <= c->ev_used; i += 4) {
            float t0f = 0, t01 = 0, t12 = 0;
            if (hipEventElapsedTime(&t0f, c->ev_pool[i], c->ev_pool[i + 1]) == hipSuccess &&
                hipEventElapsedTime(&t01, c->ev_pool[i], c->ev_pool[i + 2]) == hipSuccess &&
                hipEventElapsedTime(&t12, c->ev_pool[i + 2], c->ev_pool[i + 3]) == hipSuccess) {
                c->prof.launches++; c->rt.first_ms += t0f; c->prof.tokenize_ms += t01; c->prof.aux_ms += t12;
            }
        }
        c->ev_used = 0;
    }
    if (c->last.count_work) {  // what the general kernels counted (atomics on the control block)
        const unsigned long long *w = c->h_ctl->work;
        c->work.sentences += w[0]; c->work.B += w[1]; c->work.C += w[2]; c->work.T += w[3];
        c->work.N += w[4]; c->work.E += w[5]; c->work.K += w[6];
        for (int k = 0; k < 10; ++k) c->phase[k] += c->h_ctl->phase[k];
    }
    if (c->last.stat_slots) {  // ... and the pool kernel's wavefronts, each in its own slot
        const size_t words = (size_t)STAT_SLOTS * STAT_WORDS;
        c->stat_host.resize(words);
        HIPCHECK(hipMemcpyAsync(c->stat_host.data(), c->last.stat_slots, words * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipMemsetAsync(c->last.stat_slots, 0, words * 8, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        unsigned long long sum[STAT_WORDS] = {0};
        for (size_t i = 0; i < words; ++i) sum[i % STAT_WORDS] += c->stat_host[i];
        c->work.sentences += sum[0]; c->work.B += sum[1]; c->work.C += sum[2]; c->work.T += sum[3];
        c->work.N += sum[4]; c->work.E += sum[5]; c->work.K += sum[6];
        for (int k = 0; k < 10; ++k) c->phase[k] += sum[16 + k];
    }
    uint64_t need = c->h_ctl->n_tokens;
    if (c->last.out8 && c->h_ctl->pack_overflow) {
        if (n_tokens) *n_tokens = 0;
        set_error("a token does not fit the 8-byte record (more than 4095 chars or 262143 bytes): use the 24-byte form for this batch");
        return KGPU_ERR_CAPACITY;
    }
    if (n_tokens) *n_tokens = need;
    if (c->h_ctl->n_tokens > c->last.out_cap) {
        set_error("token buffer too small: need %llu, capacity %llu", (unsigned long long)need, (unsigned long long)c->last.out_cap);
        return KGPU_ERR_CAPACITY;
    }
    return KGPU_OK;
}

extern "C" int kgpu_ctx_set_profiling(kgpu_ctx *c, int mode) {
    if (!c) { set_error("kgpu_ctx_set_profiling: null ctx"); return KGPU_ERR_INVALID_ARG; }
    c->profiling = (mode & KGPU_PROFILE_EVENTS) != 0;
    c->event_every = (mode & KGPU_PROFILE_SAMPLED) ? 4u : 1u;
    c->launch_seq = 0;
    c->count_work = (mode & KGPU_PROFILE_WORK) != 0;
    c->count_no_t = (mode & KGPU_PROFILE_NO_T) != 0;
    return KGPU_OK;
}

extern "C" int kgpu_ctx_set_ablation(kgpu_ctx *c, int stop_after_stage) {
    if (!c || stop_after_stage < 0 || stop_after_stage > 7) { set_error("kgpu_ctx_set_ablation: bad argument"); return KGPU_ERR_INVALID_ARG; }
    c->stop_after = (uint32_t)stop_after_stage;
    return KGPU_OK;
}

extern "C" int kgpu_ctx_get_phase_cycles(kgpu_ctx *c, uint64_t out[10], int reset) {
    if (!c || !out) { set_error("kgpu_ctx_get_phase_cycles: null argument"); return KGPU_ERR_INVALID_ARG; }
    for (int k = 0; k < 10; ++k) { out[k] = c->phase[k]; if (reset) c->phase[k] = 0; }
    return KGPU_OK;
}

extern "C" int kgpu_ctx_get_work(kgpu_ctx *c, kgpu_work *out, int reset) {
    if (!c || !out) { set_error("kgpu_ctx_get_work: null argument"); return KGPU_ERR_INVALID_ARG; }
    *out = c->work;
    if (reset) c->work = kgpu_work{};
    return KGPU_OK;
}

extern "C" int kgpu_ctx_get_profile(kgpu_ctx *c, kgpu_profile *out, int reset) {
    if (!c || !out) { set_error("kgpu_ctx_get_profile: null argument"); return KGPU_ERR_INVALID_ARG; }
    *out = c->prof;
    if (reset) c->prof = kgpu_profile{};
    return KGPU_OK;
}

extern "C" int kgpu_ctx_get_plan(kgpu_ctx *c, kgpu_plan_info *out, size_t out_size) {
    if (!c || !out || out_size < 4) { set_error("kgpu_ctx_get_plan: bad argument"); return KGPU_ERR_INVALID_ARG; }
    HIPCHECK(hipSetDevice(c->dict->device));
    kgpu_plan_info p{};
    hipDeviceProp_t prop;
    p.compute_units = hipGetDeviceProperties(&prop, c->dict->device) == hipSuccess ? (uint32_t)prop.multiProcessorCount : 0u;
    if (c->plan.n_pools) {
        p.pool_lds_bytes = c->plan.pool_bytes[0]; p.pool_wavefronts = c->plan.pool_waves[0]; p.pool_max_pages = c->plan.pool_max_pages[0];
        p.pool_workgroups_per_cu = (uint32_t)pool_workgroups_per_cu(c->plan.pool_bytes[0], c->plan.pool_waves[0]);
    }
    p.window_lds_bytes = c->plan.window_lds_bytes; p.window_workgroups = (uint32_t)c->plan.window_workgroups;
    p.window_workgroups_per_cu = c->plan.window_lds_bytes ? (uint32_t)window_workgroups_per_cu(c->plan.window_lds_bytes) : 0u;
    p.streams = planned_streams();
    p.long_streams = c->own_stream ? 0u : planned_long_streams();
    p.window_first_bytes = (c->plan.n_pools && c->plan.window_lds_bytes) ? c->plan.window_first_bytes : 0u;
    std::memcpy(out, &p, std::min(out_size, sizeof p));
    return KGPU_OK;
}

extern "C" int kgpu_ctx_get_routing(kgpu_ctx *c, kgpu_routing *out, size_t out_size, int reset) {
    if (!c || !out || out_size < 8) { set_error("kgpu_ctx_get_routing: bad argument"); return KGPU_ERR_INVALID_ARG; }
    std::memcpy(out, &c->rt, std::min(out_size, sizeof(kgpu_routing)));
    if (reset) c->rt = kgpu_routing{};
    return KGPU_OK;
}

extern "C" int kgpu_dict_get_routing(kgpu_dict *d, kgpu_routing *out, size_t out_size, int reset) {
    if (!d || !out || out_size < 8) { set_error("kgpu_dict_get_routing: bad argument"); return KGPU_ERR_INVALID_ARG; }
    kgpu_routing sum{};
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        for (kgpu_ctx *c : d->pool) {
            const kgpu_routing &r = c->rt;
            sum.batches += r.batches; sum.sentences += r.sentences;
            for (int k = 0; k < 4; ++k) { sum.deferred[k] += r.deferred[k]; sum.redone[k] += r.redone[k]; }
            sum.long_launches += r.long_launches; sum.arena_regrows += r.arena_regrows; sum.first_ms += r.first_ms;
            sum.small_calls += r.small_calls; sum.small_fallbacks += r.small_fallbacks; sum.window_reruns += r.window_reruns; sum.tail_reruns += r.tail_reruns;
            sum.combined_calls += r.combined_calls; sum.combined_launches += r.combined_launches;
            if (reset) c->rt = kgpu_routing{};
        }
    }
    std::memcpy(out, &sum, std::min(out_size, sizeof sum));
    return KGPU_OK;
}

// ------------------------------------------------------- host-buffer entry point
int kgpu::pool_get(kgpu_dict *d, kgpu_ctx **out) {
    kgpu_ctx *c = nullptr;
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        if (!d->pool.empty()) { c = d->pool.back(); d->pool.pop_back(); }
    }
    if (!c) { int rc = kgpu_ctx_create(d, nullptr, &c); if (rc) return rc; }
    *out = c;
    return KGPU_OK;
}
void kgpu::pool_put(kgpu_dict *d, kgpu_ctx *c) {
    std::lock_guard<std::mutex> g(d->pool_mu);
    d->pool.push_back(c);
}

// ---- host-buffer entry point ---------------------------------------------------------------------------
// One chunk of a host call in flight on one pooled context: H2D + kernels enqueued, results still on the device.
struct HostJob {
    kgpu_ctx *c = nullptr;
    uint64_t lo = 0, m = 0;        // sentences [lo, lo + m) of the call
    std::vector<uint64_t> rel;     // chunk-relative byte offsets (must outlive the asynchronous H2D copy)
    bool active = false;
};

static int host_job_submit(HostJob &j, const uint8_t *utf8, const uint64_t *offsets) {
    kgpu_ctx *c = j.c;
    const uint64_t *off = offsets + j.lo;
    const uint64_t n = j.m, base = off[0], total = off[n] - base;
    j.rel.resize((size_t)n + 1);
    for (uint64_t i = 0; i <= n; ++i) j.rel[(size_t)i] = off[i] - base;
    const uint64_t cap = total + n + 1;  // tokens <= chars + 1 <= bytes + 1 per sentence: never too small
    int rc;
    if ((rc = c->in_utf8.ensure((size_t)total + 16)) || (rc = c->in_off.ensure((size_t)(n + 1) * 8)) ||
        (rc = c->out_tok.ensure((size_t)cap * sizeof(kgpu_token) + 64)) ||
        (rc = c->out_off.ensure((size_t)(n + 1) * 8)) || (rc = c->out_status.ensure((size_t)n + 16)))
        return rc;
    hipError_t e;
    if (total && (e = hipMemcpyAsync(c->in_utf8.p, utf8 + base, (size_t)total, hipMemcpyHostToDevice, c->stream)) != hipSuccess) { set_error("H2D utf8: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    c->h2d_queued = true;
    if ((e = hipMemcpyAsync(c->in_off.p, j.rel.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) { set_error("H2D offsets: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    if ((rc = kgpu_tokenize_device(c, (const uint8_t *)c->in_utf8.p, (const uint64_t *)c->in_off.p, n, total,
                                   (kgpu_token *)c->out_tok.p, cap, (uint64_t *)c->out_off.p, (uint8_t *)c->out_status.p)))
        return rc;
    j.active = true;
    return KGPU_OK;
}

// Wait for the job, copy its results behind the `tok_done` tokens already delivered (or only count, once the
// caller's buffer has overflowed) and make the chunk-local token offsets global.
static int host_job_finish(HostJob &j, kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets, uint8_t *status,
                           uint64_t &tok_done, bool &overflow) {
    kgpu_ctx *c = j.c;
    j.active = false;
    uint64_t got = 0;
    int rc = kgpu_ctx_sync(c, &got);
    if (rc) return rc;
    if (tok_done + got > token_capacity) overflow = true;
    hipError_t e;
    if (!overflow) {
        if (got && (e = hipMemcpyAsync(tokens + tok_done, c->out_tok.p, (size_t)got * sizeof(kgpu_token), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { set_error("D2H tokens: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
        if ((e = hipMemcpyAsync(tok_offsets + j.lo, c->out_off.p, (size_t)(j.m + 1) * 8, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { set_error("D2H offsets: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    }
    if (status && j.m && (e = hipMemcpyAsync(status + j.lo, c->out_status.p, (size_t)j.m, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { set_error("D2H status: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    // pageable destinations make these copies synchronous; pinned ones (kgpu_host_alloc) run at DMA speed while the
    // next chunks' kernels execute.  The offsets fix-up below needs the data, so wait for this stream's copies here.
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) { set_error("D2H sync: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    if (!overflow) for (uint64_t i = 0; i <= j.m; ++i) tok_offsets[j.lo + i] += tok_done;  // chunk-local -> global
    tok_done += got;
    return KGPU_OK;
}

// ---- large host calls: 8-byte records over PCIe, expanded by a few host threads ---------------------------
// What a large call moves device -> host is 24 bytes per token, seven times its input (32 tokens per 113-byte sentence on the
// cfg 2 corpus): ~65 M sentences/s at PCIe speed, and through pageable destinations far less.  Here the compaction kernel writes
// 8-byte kgpu_token8 records straight into pinned, device-mapped host memory (its stores are the transfer: no copy node, no
// D2H call), and worker threads expand them into the caller's 24-byte records (any memory: the expansion replaces the copy
// a pageable destination costs anyway) while the next chunks compute.
namespace kgpu {
// A few host threads for the large calls: expansion of the 8-byte records, staging copies.  Heap-allocated and never destroyed (its
// threads end with the process).  fork(): the child has none of the parent's threads, and its copy of the pool may hold a locked mutex
// or a condition variable with waiters -- the child handler abandons it and the first use there makes a fresh one.  Thread creation
// can fail (std::system_error): start() reports how many threads run, and with none the caller takes the path that needs none.
unsigned WorkerPool::start() {
    std::lock_guard<std::mutex> g(mu);
    if (!th.empty()) return (unsigned)th.size();
    unsigned n = 0;
    if (const char *e = getenv("KGPU_HOST_THREADS")) n = (unsigned)atoi(e);
    if (n == 0) n = std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 8));
    for (unsigned i = 0; i < n; ++i) {
        try {
            th.emplace_back([this] {
                for (;;) {
                    std::function<void()> f;
                    { std::unique_lock<std::mutex> l(mu); cv.wait(l, [this] { return !q.empty(); }); f = std::move(q.front()); q.pop_front(); }
                    f();
                }
            });
        } catch (...) { break; }  // out of threads: run with what there is
    }
    return (unsigned)th.size();
}
void WorkerPool::submit(std::function<void()> f) { { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(f)); } cv.notify_one(); }
// A counter of tasks reaching zero: a short spin (the common case: the workers are almost through), then sleeps on the pool's
// completion signal instead of burning a core the workers could use.
void WorkerPool::wait_zero(std::atomic<int> &counter) {
    // (the tasks waited for here are tens of microseconds long: a sleep costs more than it saves until the wait has lasted a while)
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        for (int spin = 0; spin < 64; ++spin) { if (counter.load(std::memory_order_acquire) == 0) return; std::this_thread::yield(); }
        timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000 > 2000) break;
    }
    std::unique_lock<std::mutex> l(done_mu);
    while (counter.load(std::memory_order_acquire) != 0) done_cv.wait_for(l, std::chrono::microseconds(200));
}
void WorkerPool::task_done(std::atomic<int> &counter) {
    if (counter.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> g(done_mu); done_cv.notify_all(); }
}
static std::atomic<WorkerPool *> g_pool{nullptr};
static std::atomic<int> g_pool_lock{0};
static void pool_atfork_child() { g_pool.store(nullptr, std::memory_order_relaxed); g_pool_lock.store(0, std::memory_order_relaxed); }
WorkerPool &workers() {
    WorkerPool *w = g_pool.load(std::memory_order_acquire);
    if (w) return *w;
    while (g_pool_lock.exchange(1, std::memory_order_acquire)) std::this_thread::yield();
    static bool hooked = false;
    if (!hooked) { hooked = true; pthread_atfork(nullptr, nullptr, pool_atfork_child); }
    w = g_pool.load(std::memory_order_relaxed);
    if (!w) { w = new WorkerPool(); g_pool.store(w, std::memory_order_release); }
    g_pool_lock.store(0, std::memory_order_release);
    return *w;
}
}  // namespace kgpu

struct PipeJob {
    kgpu_ctx *c = nullptr;
    uint64_t lo = 0, m = 0, total = 0, cap = 0;
    size_t off_first = 0, off_toff = 0, off_status = 0;  // inside pin_out: records | first | token offsets | status
    std::atomic<int> tasks{0};                          // expansion tasks still reading pin_out
    bool active = false;
    bool stream = false;                                // the call is large: its 24-byte records are written with non-temporal stores (kgpu_runtime.h: expand_tokens)
};

// memcpy of a large block with the workers' help (the calling thread's staging copy is what limits a large call otherwise)
void kgpu::parallel_copy(void *dst, const void *src, size_t bytes) {
    constexpr size_t PIECE = 256 * 1024;
    if (bytes < 2 * PIECE) { std::memcpy(dst, src, bytes); return; }
    const size_t np = std::min<size_t>(8, bytes / PIECE), each = ((bytes + np - 1) / np + 63) & ~(size_t)63;  // np * each >= bytes (rounded UP: a floor here lost the last bytes of a chunk)
    std::atomic<int> left{(int)np - 1};
    for (size_t k = 1; k < np; ++k) {
        const size_t lo = k * each, hi = std::min(bytes, lo + each);
        std::atomic<int> *l = &left;
        workers().submit([=] { if (hi > lo) std::memcpy((uint8_t *)dst + lo, (const uint8_t *)src + lo, hi - lo); workers().task_done(*l); });
    }
    std::memcpy(dst, src, std::min(bytes, each));
    workers().wait_zero(left);
}

bool kgpu::is_pinned_host(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// The chunk's input goes to the device as ONE block [offsets (absolute, as the caller has them) | bytes]; the kernels subtract
// offsets[0] themselves, the text pointer is biased by it.  Pinned caller memory is copied from directly (DMA), pageable memory through
// the context's pinned staging block, filled with the workers' help.
static int pipe_submit(PipeJob &j, const uint8_t *utf8, const uint64_t *offsets, bool pinned_in) {
    kgpu_ctx *c = j.c;
    workers().wait_zero(j.tasks);  // the block's previous results are still being expanded
    const uint64_t *off = offsets + j.lo;
    const uint64_t n = j.m, base = off[0], total = off[n] - base;
    j.total = total;
    j.cap = total + n + 1;  // tokens <= chars + 1 <= bytes + 1 per sentence: never too small
    const size_t in_off_bytes = ((size_t)(n + 1) * 8 + 63) & ~(size_t)63, in_bytes = in_off_bytes + (size_t)total + 16;
    j.off_first = ((size_t)j.cap * 8 + 63) & ~(size_t)63;
    j.off_toff = j.off_first + (((size_t)n * 8 + 63) & ~(size_t)63);
    j.off_status = j.off_toff + (((size_t)(n + 1) * 8 + 63) & ~(size_t)63);
    int rc;
    if ((rc = c->in_block.ensure(in_bytes)) || (rc = c->pin_out.ensure(j.off_status + (size_t)n + 64, true)) || (rc = c->out_status.ensure((size_t)n + 16)) || (rc = c->out_off.ensure((size_t)(n + 1) * 8)) ||
        (!pinned_in && (rc = c->pin_in.ensure(in_bytes, false))))
        return rc;
    hipError_t e;
    uint8_t *dblk = (uint8_t *)c->in_block.p;
    if (pinned_in) {
        c->h2d_queued = true;
        if ((e = hipMemcpyAsync(dblk, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
            (total && (e = hipMemcpyAsync(dblk + in_off_bytes, utf8 + base, (size_t)total, hipMemcpyHostToDevice, c->stream)) != hipSuccess)) {
            set_error("H2D input: %s", hipGetErrorString(e)); return KGPU_ERR_HIP;
        }
    } else {
        std::memcpy(c->pin_in.h, off, (size_t)(n + 1) * 8);
        if (total) parallel_copy((uint8_t *)c->pin_in.h + in_off_bytes, utf8 + base, (size_t)total);
        c->h2d_queued = true;
        if ((e = hipMemcpyAsync(dblk, c->pin_in.h, in_off_bytes + (size_t)total, hipMemcpyHostToDevice, c->stream)) != hipSuccess) { set_error("H2D input block: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    }
    uint8_t *po = (uint8_t *)c->pin_out.d;
    if ((rc = tokenize_device_impl(c, dblk + in_off_bytes - base, (const uint64_t *)dblk, n, total, nullptr,
                                   (kgpu_token8 *)po, (uint32_t *)(po + j.off_first), po + j.off_status, (uint64_t *)(po + j.off_toff), j.cap,
                                   (uint64_t *)c->out_off.p, (uint8_t *)c->out_status.p, "kgpu_tokenize_batch")))
        return rc;
    j.active = true;
    return KGPU_OK;
}

static int host_job_submit(struct HostJob &j, const uint8_t *utf8, const uint64_t *offsets);
static int host_job_finish(struct HostJob &j, kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets, uint8_t *status, uint64_t &tok_done, bool &overflow);

// Wait for the chunk's kernels (its records are in host memory then), hand the expansion to the workers in slices of 2048 sentences.
static int pipe_finish(PipeJob &j, const uint8_t *utf8, const uint64_t *offsets, kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets,
                       uint8_t *status, uint64_t &tok_done, bool &overflow, std::atomic<int> &outstanding) {
    kgpu_ctx *c = j.c;
    j.active = false;
    uint64_t got = 0;
    int rc = kgpu_ctx_sync(c, &got);
    if (rc == KGPU_ERR_CAPACITY && c->h_ctl->pack_overflow) {  // a token beyond the 8-byte packing: this chunk once more, 24-byte records, the plain way
        // (the previous chunk's last expansion slice stores tok_offsets[j.lo] too -- the boundary entry -- and host_job_finish copies and then adds
        // to it: wait for the expansions in flight first)
        workers().wait_zero(outstanding);
        HostJob hj;
        hj.c = c; hj.lo = j.lo; hj.m = j.m;
        if ((rc = host_job_submit(hj, utf8, offsets))) return rc;
        return host_job_finish(hj, tokens, token_capacity, tok_offsets, status, tok_done, overflow);
    }
    if (rc) return rc;
    if (tok_done + got > token_capacity) overflow = true;
    const uint64_t tok_base = tok_done;
    tok_done += got;
    const uint8_t *ph = (const uint8_t *)c->pin_out.h;
    const kgpu_token8 *rec = (const kgpu_token8 *)ph;
    const uint32_t *first = (const uint32_t *)(ph + j.off_first);
    const uint64_t *toff = (const uint64_t *)(ph + j.off_toff);
    const uint8_t *st = ph + j.off_status;
    const uint64_t SLICE = std::min<uint64_t>(2048, std::max<uint64_t>(256, j.m / 8));  // (a lone 4096-sentence call: eight slices, not two)
    const int nt = (int)((j.m + SLICE - 1) / SLICE);
    if (nt == 0) { if (!overflow) tok_offsets[j.lo] = tok_base; return KGPU_OK; }
    j.tasks.store(nt, std::memory_order_release);
    outstanding.fetch_add(nt, std::memory_order_acq_rel);
    const bool ovf = overflow, jstream = j.stream;
    const uint64_t lo = j.lo, m = j.m;
    for (int t = 0; t < nt; ++t) {
        const uint64_t a = (uint64_t)t * SLICE, b = std::min(m, a + SLICE);
        std::atomic<int> *jt = &j.tasks, *out = &outstanding;
        workers().submit([=] {
            if (!ovf) {
                const bool stream = jstream;
                expand_tokens(rec + toff[a], toff + a, first + 2 * a, b - a, tokens + tok_base + toff[a], stream);
                for (uint64_t i = a; i < b; ++i) tok_offsets[lo + i] = tok_base + toff[i];
                if (b == m) tok_offsets[lo + m] = tok_base + toff[m];
                if (stream) expand_fence();
            }
            if (status) std::memcpy(status + lo + a, st + a, (size_t)(b - a));
            workers().task_done(*jt);
            workers().task_done(*out);
        });
    }
    return KGPU_OK;
}

// ---- small calls: ONE launch, no copies ------------------------------------------------------------------
// The reference's call shape is one sentence per call (src/bin/kanpyo.rs:106-126); the general path costs such a call
// five dependent launches, two host-to-device and three device-to-host copies (~120 us).  Here the sentences and the
// results live in one pinned, device-mapped block: the pool kernel reads its input over PCIe, tokenizes one sentence
// per wavefront, compacts and publishes by itself (kgpu_pool.hip, `fused_host`), and the host polls a sequence number.
static constexpr uint64_t SMALL_MAX_N = 128, SMALL_MAX_BYTES = 16 * 1024;
static constexpr size_t SM_OFF_OFFS = SMALL_MAX_BYTES + 64, SM_OFF_TOK = SM_OFF_OFFS + (SMALL_MAX_N + 1) * 8 + 56,
                        SM_OFF_TOFF = SM_OFF_TOK + (SMALL_MAX_BYTES + SMALL_MAX_N) * sizeof(kgpu_token),
                        SM_OFF_STATUS = SM_OFF_TOFF + (SMALL_MAX_N + 1) * 8 + 56, SM_BYTES = SM_OFF_STATUS + SMALL_MAX_N + 64;

// One caller's part of a single-launch small call.
struct SmallReq {
    const uint8_t *utf8; const uint64_t *offsets; uint64_t n;
    kgpu_token *tokens; uint64_t token_capacity; uint64_t *tok_offsets; uint8_t *status; uint64_t *n_tokens;
    int rc = -1;              // KGPU_OK: done; KGPU_ERR_CAPACITY: done, this caller's buffer too small; -1: not served here (a sentence needs the long way)
    char err[160] = "";       // the message behind rc (set_error is thread-local: the caller's thread repeats it)
    bool done = false;
};

// The launch for one or more callers' sentences (`reqs` in arrival order; together at most SMALL_MAX_N sentences / SMALL_MAX_BYTES).
// Returns KGPU_OK when the launch itself went through (every request then has its own rc), else the error (no request was served).
// KGPU_SMALL_TRACE=1: where a small call's wall time goes, summed over the process (microseconds): kgpu_debug_small_trace reads and resets
static std::atomic<uint64_t> g_st[8];  // launches, prep ns, launch-call ns, poll ns, hand-out ns, sentences
static inline uint64_t now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }
extern "C" void kgpu_debug_small_trace(uint64_t out[8]) { for (int k = 0; k < 8; ++k) out[k] = g_st[k].exchange(0); }
// ... and where the callers' CPU time goes (CLOCK_THREAD_CPUTIME_ID at the phase boundaries, ns summed over all threads): [0] calls that joined a batch,
// [1] calls that led one, [2] entry + the combiner's lock, [3] a follower's wait, [4] the leader's window, [5] close + context, [6] assembling the launch,
// [7] the launch call, [8] the poll, [9] handing the records out, [10] waking the followers, [11] hipSetDevice at the entry point
static std::atomic<uint64_t> g_sc[16];
static inline uint64_t cpu_ns() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }
static bool small_trace_on() { static const bool on = env_flag_now("KGPU_SMALL_TRACE"); return on; }
extern "C" void kgpu_debug_small_cpu(uint64_t out[16]) { for (int k = 0; k < 16; ++k) out[k] = g_sc[k].exchange(0); }

static unsigned cpu_budget();
static void short_sleep_us(unsigned us);
static int small_call(kgpu_dict *d, kgpu_ctx *c, SmallReq *const *reqs, size_t nreq) {
    static const bool trace = env_flag_now("KGPU_SMALL_TRACE");
    const uint64_t tt0 = trace ? now_ns() : 0, cc0 = trace ? cpu_ns() : 0;
    uint64_t n = 0, total = 0;
    for (size_t r = 0; r < nreq; ++r) { n += reqs[r]->n; total += reqs[r]->offsets[reqs[r]->n] - reqs[r]->offsets[0]; }
    int rc;
    if (!c->sm_host) {
        if (hipHostMalloc((void **)&c->sm_host, SM_BYTES, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&c->sm_dev, c->sm_host, 0) != hipSuccess) {
            if (c->sm_host) { (void)hipHostFree(c->sm_host); c->sm_host = nullptr; }
            (void)hipGetLastError();
            return KGPU_OK;  // every request keeps rc = -1: the general path
        }
    }
    if (c->pending && (rc = kgpu_ctx_sync(c, nullptr)) != KGPU_OK && rc != KGPU_ERR_CAPACITY) return rc;
    // (no scratch arena: this path launches the pool kernel alone, whose lattices live in LDS -- a context that only ever serves small calls holds no 256 MiB)
    if ((rc = c->stage.ensure((size_t)(total + n + 1) * sizeof(kgpu_token) + 64)) ||
        (rc = c->tok_count.ensure((size_t)(n + 1) * 4)) || (rc = c->ovf.ensure((size_t)(n + 1) * 4 * 4)))
        return rc;
    uint64_t *h_off = (uint64_t *)(c->sm_host + SM_OFF_OFFS);
    {
        uint64_t at = 0, si = 0;
        for (size_t r = 0; r < nreq; ++r) {
            const SmallReq &q = *reqs[r];
            const uint64_t base = q.offsets[0], bytes = q.offsets[q.n] - base;
            if (bytes) std::memcpy(c->sm_host + at, q.utf8 + base, (size_t)bytes);
            for (uint64_t i = 0; i < q.n; ++i) h_off[si + i] = at + (q.offsets[i] - base);
            at += bytes; si += q.n;
        }
        h_off[n] = at;
    }
    const uint32_t seq = ++c->sm_seq ? c->sm_seq : ++c->sm_seq;  // never 0
    __atomic_store_n(&c->h_ctl->small_flag, 0u, __ATOMIC_RELEASE);
    BatchArgs a{};
    a.utf8 = c->sm_dev; a.offsets = (const uint64_t *)(c->sm_dev + SM_OFF_OFFS); a.n = n; a.ctl = c->d_ctl;
    a.arena = (uint8_t *)c->arena.p; a.arena_bytes = c->arena.bytes;
    a.stage = (kgpu_token *)c->stage.p; a.tok_count = (uint32_t *)c->tok_count.p;
    a.status = c->sm_dev + SM_OFF_STATUS; a.out = (kgpu_token *)(c->sm_dev + SM_OFF_TOK); a.out_cap = total + n;
    a.tok_offsets = (uint64_t *)(c->sm_dev + SM_OFF_TOFF);
    a.est_q8 = d->est_q8.load(std::memory_order_relaxed);
    for (int k = 0; k < 4; ++k) a.ovf[k] = (uint32_t *)c->ovf.p + (size_t)k * (n + 1);
    a.fused_host = c->h_ctl_dev; a.fused_seq = seq;
    if (c->ctl_dirty) HIPCHECK(hipMemsetAsync(c->d_ctl, 0, sizeof(Control), c->stream));
    c->ctl_dirty = true;
    const uint64_t tt1 = trace ? now_ns() : 0, cc1 = trace ? cpu_ns() : 0;
    {
        hipError_t e = (hipError_t)launch_small_call(d->view, a, c->plan, c->stream);
        if (e != hipSuccess) { set_error("small-call launch: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    }
    const uint64_t tt2 = trace ? now_ns() : 0, cc2 = trace ? cpu_ns() : 0;
    // poll the sequence number (the kernel's last store); a stream query now and then catches a failed launch
    for (uint64_t spin = 0;; ++spin) {
        if (__atomic_load_n(&c->h_ctl->small_flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spin & 0xFFFF) == 0xFFFF) {
            hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(&c->h_ctl->small_flag, __ATOMIC_ACQUIRE) == seq) break;
                set_error("small call: the kernel finished without publishing"); return KGPU_ERR_INTERNAL;
            }
            if (q != hipErrorNotReady) { set_error("small call: %s", hipGetErrorString(q)); return KGPU_ERR_HIP; }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        // many callers inside the entry point: most of them need a CPU to assemble or pick up their results -- this thread's poll lets them have it;
        // more callers than CPUs: sleep through most of the launch's ~45 us instead (the poll costs the group's CPU quota, the sleep does not)
        if ((spin & 63) == 63) {
            const int callers = d->combiner_callers();
            if (callers > (int)cpu_budget()) short_sleep_us(spin < 64 ? 25 : 8);
            else if (callers > 8) sched_yield();
        }
    }
    const uint64_t tt3 = trace ? now_ns() : 0, cc3 = trace ? cpu_ns() : 0;
    struct TraceOut { bool on; uint64_t t0, t1, t2, t3, n, c0, c1, c2, c3; ~TraceOut() { if (on) { const uint64_t t4 = now_ns(), c4 = cpu_ns(); g_st[0] += 1; g_st[1] += t1 - t0; g_st[2] += t2 - t1; g_st[3] += t3 - t2; g_st[4] += t4 - t3; g_st[5] += n;
        g_sc[6] += c1 - c0; g_sc[7] += c2 - c1; g_sc[8] += c3 - c2; g_sc[9] += c4 - c3; } } } trace_out{trace, tt0, tt1, tt2, tt3, n, cc0, cc1, cc2, cc3};
    c->ctl_dirty = false;  // the publishing wavefront zeroed the device block
    c->rt.batches++; c->rt.sentences += n;
    c->rt.deferred[0] += c->h_ctl->ovf_count[0]; c->rt.redone[0] += c->h_ctl->late_count[0];
    c->rt.small_calls += nreq;
    if (nreq > 1) { c->rt.combined_calls += nreq; c->rt.combined_launches++; }
    if (c->h_ctl->ovf_count[0] != 0 || c->h_ctl->arena_overflow || c->h_ctl->small_abort) {  // a sentence left for the long / HBM-scratch kernels, or the rendezvous timed out
        c->rt.small_fallbacks += nreq;
        return KGPU_OK;  // rc = -1 everywhere: each caller takes the general path with its own sentences
    }
    const uint64_t *h_toff = (const uint64_t *)(c->sm_host + SM_OFF_TOFF);
    const kgpu_token *h_tok = (const kgpu_token *)(c->sm_host + SM_OFF_TOK);
    uint64_t si = 0;
    for (size_t r = 0; r < nreq; ++r) {   // every caller's dense slice
        SmallReq &q = *reqs[r];
        const uint64_t t0 = h_toff[si], got = h_toff[si + q.n] - t0;
        if (q.n_tokens) *q.n_tokens = got;
        if (got > q.token_capacity) {
            snprintf(q.err, sizeof q.err, "token buffer too small: need %llu, capacity %llu", (unsigned long long)got, (unsigned long long)q.token_capacity);
            q.rc = KGPU_ERR_CAPACITY;
        } else {
            if (got) std::memcpy(q.tokens, h_tok + t0, (size_t)got * sizeof(kgpu_token));
            for (uint64_t i = 0; i <= q.n; ++i) q.tok_offsets[i] = h_toff[si + i] - t0;
            if (q.status) std::memcpy(q.status, c->sm_host + SM_OFF_STATUS + si, (size_t)q.n);
            q.rc = KGPU_OK;
        }
        si += q.n;
    }
    return KGPU_OK;
}

// ---- the combiner: concurrent small calls share one launch -------------------------------------------------------------
// The reference's tokenize() takes &self and is Send + Sync (src/tokenizer.rs:16): a server calls it from many threads, one sentence
// per call (src/bin/kanpyo.rs:106-126).  A launch costs the same ~50 us whether it carries one sentence or a hundred, so callers that
// arrive while another small call is being assembled join it: the first one in is the leader -- it keeps the batch open for a short
// window (only while other callers are inside the entry point: a lone caller never waits), takes a pooled context, launches, and hands
// every follower its own dense slice back.  Followers sleep on a condition variable meanwhile.
// The combiner's lock: held for a push_back and two additions (tens of nanoseconds), taken by every caller -- and by a whole batch's followers at the same
// instant, when the leader's one wake-up releases them into their next calls.  A pthread mutex puts each of them to sleep and wakes it again through the kernel:
// measured with 128 callers, 40-48 us of (system) CPU per call in the lock alone -- more CPU than a 16-CPU cgroup quota grants, so the group spent most of each
// 100 ms period throttled (profiles/experiments/r05_callers_cpu.txt).  Test-and-test-and-set with pause and backoff, a few yields, then asleep on the word.
struct SpinLock {
    // 0 free, 1 held, 2 held and somebody may be asleep on the word.  Spinning is bounded (round 6, advisor): a holder that lost its CPU -- more callers than
    // CPUs, a cgroup quota, a lower-priority holder under SCHED_FIFO -- is not waited for with yields for ever; after ~four yields the waiter sleeps on the
    // word (futex) and the release wakes one sleeper.  The uncontended and the briefly contended paths never enter the kernel.
    std::atomic<uint32_t> v{0};
    void lock() {
        // (a lost compare-and-swap backs off for twice as long, up to 32 pauses: two dozen threads that all saw the word free do not all write it again at the next release)
        for (unsigned spins = 0, backoff = 1, yields = 0; yields < 4;) {
            if (v.load(std::memory_order_relaxed) == 0) {
                uint32_t z = 0;
                if (v.compare_exchange_weak(z, 1, std::memory_order_acquire, std::memory_order_relaxed)) return;
                for (unsigned k = 0; k < backoff; ++k) cpu_relax();
                if (backoff < 32) backoff *= 2;
            }
            cpu_relax();
            if (++spins >= 2048) { sched_yield(); spins = 0; ++yields; }
        }
        while (v.exchange(2, std::memory_order_acquire) != 0) syscall(SYS_futex, (uint32_t *)&v, FUTEX_WAIT_PRIVATE, 2u, nullptr, nullptr, 0);
    }
    void unlock() {
        if (v.exchange(0, std::memory_order_release) == 2) syscall(SYS_futex, (uint32_t *)&v, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    }
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
};
struct Combiner {
    // A batch lives on the heap, shared by its leader and its followers (a follower may still be reading its own result when the leader returns).
    struct Batch {
        std::vector<SmallReq *> reqs; uint64_t n = 0, bytes = 0; bool closed = false;
        std::atomic<uint32_t> done{0};   // futex word: followers sleep on it, ONE wake-all syscall releases them (no shared condition variable: a
    };                                   // batch's completion wakes its own followers only, and nobody queues on a mutex to find out)
    // (each on a cache line of its own: every caller adds itself to `callers` on the way in and out, the lock's waiters read the lock word meanwhile)
    alignas(64) SpinLock mu;
    std::shared_ptr<Batch> open;
    alignas(64) std::atomic<int> callers{0};     // threads inside the small-call entry
    alignas(64) std::atomic<int> in_flight{0};   // launches between close and completion
};
int kgpu_dict::combiner_callers() const { return combiner ? combiner->callers.load(std::memory_order_relaxed) : 0; }
static Combiner *combiner_new() { return new Combiner(); }
static void combiner_delete(Combiner *c) { delete c; }
static Combiner &combiner_of(kgpu_dict *d) { return *d->combiner; }
static unsigned combine_window_us() {
    static const unsigned us = [] { const char *e = getenv("KGPU_COMBINE_US"); const int v = e ? atoi(e) : 12; return (unsigned)(v < 0 ? 0 : v > 1000 ? 1000 : v); }();
    return us;
}
static int combine_max_in_flight() {
    static const int v = [] { const char *e = getenv("KGPU_COMBINE_LAUNCHES"); const int x = e ? atoi(e) : 4; return x < 1 ? 1 : x > 64 ? 64 : x; }();
    return v;
}
// CPUs this process may actually use: hardware threads, narrowed by the affinity mask and the cgroup's CPU quota (cpu.max, or cfs_quota_us / cfs_period_us).
// With more callers inside the entry point than that, a spinning thread takes the CPU a sleeping caller needs -- and under a CFS quota the spinning burns the
// whole group's budget for the period (round 4: 128 threads on a 16-CPU quota, p99 64 ms) -- so the waits below sleep instead of spinning.
static unsigned cpu_budget() {
    static const unsigned n = [] {
        unsigned hw = std::thread::hardware_concurrency();
        if (!hw) hw = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const unsigned a = (unsigned)CPU_COUNT(&set); if (a && a < hw) hw = a; }
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32];
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
            if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
        }
        if (quota > 0 && period > 0) { const unsigned c = (unsigned)((quota + period - 1) / period); if (c && c < hw) hw = c; }
        if (const char *e = getenv("KGPU_CPU_BUDGET")) { const int v = atoi(e); if (v > 0) hw = (unsigned)v; }   // (tests: force the crowded mode)
        return hw;
    }();
    return n;
}
// A short sleep (the default 50 us timer slack would make 15 us into 65: the slack is lowered for the sleep and put back)
static void short_sleep_us(unsigned us) {
    const int slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
    if (slack > 2000) prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    timespec ts{0, (long)us * 1000};
    nanosleep(&ts, nullptr);
    if (slack > 2000) prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
}
static void futex_wait(std::atomic<uint32_t> *w, uint32_t expected) { syscall(SYS_futex, (uint32_t *)w, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0); }
static void futex_wake_all(std::atomic<uint32_t> *w) { syscall(SYS_futex, (uint32_t *)w, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

// KGPU_OK / KGPU_ERR_CAPACITY: served; -1: take the general path; other: error
static int small_call_combined(kgpu_dict *d, SmallReq &me) {
    Combiner &cb = combiner_of(d);
    const uint64_t my_bytes = me.offsets[me.n] - me.offsets[0];
    struct CallerCount { std::atomic<int> &c; CallerCount(std::atomic<int> &c_) : c(c_) { c.fetch_add(1, std::memory_order_acq_rel); } ~CallerCount() { c.fetch_sub(1, std::memory_order_acq_rel); } } in(cb.callers);
    std::shared_ptr<Combiner::Batch> mine;
    const bool trace = small_trace_on();
    uint64_t k0 = trace ? cpu_ns() : 0;
    for (;;) {   // (a batch is allocated OUTSIDE the lock -- the lock is held for a push_back and two additions -- and only by a caller that found none to join)
        {
            std::unique_lock<SpinLock> l(cb.mu);
            std::shared_ptr<Combiner::Batch> b = cb.open;
            if (b && !b->closed && b->n + me.n <= SMALL_MAX_N && b->bytes + my_bytes <= SMALL_MAX_BYTES) {   // join the batch being assembled
                b->reqs.push_back(&me); b->n += me.n; b->bytes += my_bytes;
                l.unlock();
                const uint64_t k1 = trace ? cpu_ns() : 0;
                while (b->done.load(std::memory_order_acquire) == 0) futex_wait(&b->done, 0);   // the leader has written my records and my rc before it sets the word
                if (trace) { g_sc[0] += 1; g_sc[2] += k1 - k0; g_sc[3] += cpu_ns() - k1; }
                if (me.rc > 0 && me.err[0]) set_error("%s", me.err);
                return me.rc;
            }
            if (mine) {
                mine->reqs.push_back(&me); mine->n = me.n; mine->bytes = my_bytes;
                cb.open = mine;   // (a batch another leader still holds open but that has no room for me stays its leader's: it closes it itself)
                break;
            }
        }
        mine = std::make_shared<Combiner::Batch>();
        mine->reqs.reserve(SMALL_MAX_N);   // (no reallocation under the lock later)
    }
    // Leader.  A lone caller launches at once.  With other callers inside the entry point the batch stays open for a short window -- and, when
    // the device already has its fill of small launches in flight, until one of them completes (or the batch is full): the batch size follows the
    // load (group commit), the number of launches per second stays what the streams carry.  Spinning: the waits are shorter than a futex sleep.
    if (trace) { const uint64_t k1 = cpu_ns(); g_sc[1] += 1; g_sc[2] += k1 - k0; k0 = k1; }
    const unsigned win = combine_window_us();
    if (win && cb.callers.load(std::memory_order_acquire) > 1) {
        timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            if (cb.callers.load(std::memory_order_relaxed) > (int)cpu_budget()) short_sleep_us(10);   // more callers than CPUs: the window is slept, not spun
            else for (int k = 0; k < 16; ++k) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            const long long us = (t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000;
            const bool busy = cb.in_flight.load(std::memory_order_acquire) >= combine_max_in_flight();
            if ((us >= (long long)win && !busy) || us >= 400) break;   // (a four times longer window when callers exceed CPUs: measured, 64 threads 358 -> 301 k sentences/s, 128 unchanged)
            std::lock_guard<SpinLock> g(cb.mu);
            if (mine->n >= SMALL_MAX_N || mine->bytes + 256 > SMALL_MAX_BYTES) break;  // full
            if (!busy && (int)mine->reqs.size() >= cb.callers.load(std::memory_order_acquire)) break;  // everyone who is here is in
        }
    }
    if (trace) { const uint64_t k1 = cpu_ns(); g_sc[4] += k1 - k0; k0 = k1; }
    {
        std::lock_guard<SpinLock> g(cb.mu);
        mine->closed = true;
        if (cb.open == mine) cb.open.reset();
    }
    cb.in_flight.fetch_add(1, std::memory_order_acq_rel);
    kgpu_ctx *c = nullptr;
    int rc = pool_get(d, &c);
    if (trace) { const uint64_t k1 = cpu_ns(); g_sc[5] += k1 - k0; k0 = k1; }
    if (!rc) {
        rc = c->plan.n_pools ? small_call(d, c, mine->reqs.data(), mine->reqs.size()) : KGPU_OK;
        pool_put(d, c);
    }
    cb.in_flight.fetch_sub(1, std::memory_order_acq_rel);
    const int my_rc = rc ? rc : me.rc;
    if (rc) for (SmallReq *q : mine->reqs) { q->rc = rc; snprintf(q->err, sizeof q->err, "%s", kgpu_last_error()); }
    const bool had_followers = mine->reqs.size() > 1;
    if (trace) k0 = cpu_ns();
    mine->done.store(1, std::memory_order_release);   // (followers may return -- and their SmallReq die -- from here on: nothing of theirs is touched below)
    if (had_followers) futex_wake_all(&mine->done);
    if (trace) g_sc[10] += cpu_ns() - k0;
    if (my_rc > 0 && !rc && me.err[0]) set_error("%s", me.err);
    return my_rc;
}

// The plain form of a large call (24-byte records, device-to-host copies on the context's stream): what a chunk falls back to when
// a token does not fit the 8-byte packing, and the whole call under KGPU_HOST_LEGACY (tests: both forms give the same records).
static int tokenize_batch_legacy(kgpu_dict *d, const uint8_t *utf8, const uint64_t *offsets, uint64_t n, kgpu_token *tokens, uint64_t token_capacity,
                                 uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens, const TestHooks &hooks) {
    // A large call goes through in chunks (bounded device staging: 24 B per input byte), three of them in
    // flight on pooled contexts: while chunk k's results travel to the host, chunk k+1's kernels run and
    // chunk k+2's input is on its way.  Results are delivered in order, so the tokens stay dense.
    const uint64_t CHUNK_BYTES = hooks.chunk_bytes, CHUNK_SENTS = hooks.chunk_sents;
    constexpr int DEPTH = 4;
    HostJob jobs[DEPTH];
    int rc = KGPU_OK, njobs = 0;
    uint64_t done = 0, tok_done = 0;
    bool overflow = false;
    tok_offsets[0] = 0;
    int head = 0, inflight = 0;  // jobs[head .. head + inflight) (mod DEPTH) are active, oldest first
    while (!rc && (done < n || (n == 0 && done == 0 && inflight == 0))) {
        if (inflight == DEPTH) {
            rc = host_job_finish(jobs[head], tokens, token_capacity, tok_offsets, status, tok_done, overflow);
            head = (head + 1) % DEPTH; --inflight;
            if (rc) break;
        }
        HostJob &j = jobs[(head + inflight) % DEPTH];
        if (!j.c) {
            {
                std::lock_guard<std::mutex> g(d->pool_mu);
                if (!d->pool.empty()) { j.c = d->pool.back(); d->pool.pop_back(); }
            }
            if (!j.c && (rc = kgpu_ctx_create(d, nullptr, &j.c))) break;
            ++njobs;
        }
        uint64_t m = 0;
        while (done + m < n && m < CHUNK_SENTS && (m == 0 || offsets[done + m + 1] - offsets[done] <= CHUNK_BYTES)) ++m;
        j.lo = done; j.m = m;
        if ((rc = host_job_submit(j, utf8, offsets))) break;
        ++inflight;
        done += m;
        if (n == 0) break;
    }
    while (inflight) {  // drain in order (also after an error: the contexts go back to the pool idle)
        int r2 = host_job_finish(jobs[head], tokens, token_capacity, tok_offsets, status, tok_done, overflow);
        if (!rc) rc = r2;
        head = (head + 1) % DEPTH; --inflight;
    }
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        for (int k = 0; k < DEPTH; ++k)
            if (jobs[k].c) d->pool.push_back(jobs[k].c);
    }
    (void)njobs;
    if (n_tokens) *n_tokens = tok_done;
    if (!rc && overflow) {
        set_error("token buffer too small: need %llu, capacity %llu", (unsigned long long)tok_done, (unsigned long long)token_capacity);
        return KGPU_ERR_CAPACITY;
    }
    return rc;
}

extern "C" int kgpu_tokenize_batch(kgpu_dict *d, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                                   kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets,
                                   uint8_t *status, uint64_t *n_tokens) {
    if (!d || !offsets || !tok_offsets || (token_capacity && !tokens)) {
        set_error("kgpu_tokenize_batch: null argument");
        return KGPU_ERR_INVALID_ARG;
    }
    for (uint64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) { set_error("kgpu_tokenize_batch: offsets not monotone at %llu", (unsigned long long)i); return KGPU_ERR_INVALID_ARG; }
    if (offsets[n] - offsets[0] && !utf8) { set_error("kgpu_tokenize_batch: null utf8"); return KGPU_ERR_INVALID_ARG; }
    const uint64_t kd0 = small_trace_on() ? cpu_ns() : 0;
    HIPCHECK(hipSetDevice(d->device));
    if (kd0) g_sc[11] += cpu_ns() - kd0;   // (KGPU_SMALL_TRACE: the CPU time of hipSetDevice)

    if (n >= 1 && n <= SMALL_MAX_N && offsets[n] - offsets[0] <= SMALL_MAX_BYTES && !test_hooks().no_small_calls) {
        SmallReq me{utf8, offsets, n, tokens, token_capacity, tok_offsets, status, n_tokens};
        const int rc = small_call_combined(d, me);
        if (rc != -1) return rc;  // -1: a sentence needs a kernel this path does not launch: take the general path below
    }

    // A large call goes through in chunks on pooled contexts, several in flight: while chunk k's records are expanded on the host,
    // chunk k+1 .. k+4 compute and chunk k+5's input is on its way.  Results are delivered in order, so the tokens stay dense.
    const TestHooks hooks = test_hooks();
    if (hooks.legacy_host_path) return tokenize_batch_legacy(d, utf8, offsets, n, tokens, token_capacity, tok_offsets, status, n_tokens, hooks);
    if (workers().start() == 0) return tokenize_batch_legacy(d, utf8, offsets, n, tokens, token_capacity, tok_offsets, status, n_tokens, hooks);  // no worker threads to be had
    static const bool trace = env_flag_now("KGPU_HOST_TRACE");  // where the calling thread's time goes, per call, on stderr
    double t_submit = 0, t_finish = 0, t_tail = 0;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    const double t_call = now();
    // chunks of 8192 sentences, ten in the device pipeline (measured on 400k sentences: 16384 x 6: 55.6, 8192 x 10: 61.2, 4096 x 14: 58.8 M sentences/s)
    const uint64_t CHUNK_BYTES = std::min<uint64_t>(hooks.chunk_bytes, 2ull << 20);
    const uint64_t CHUNK_SENTS = std::min<uint64_t>(hooks.chunk_sents, std::min<uint64_t>(8192, std::max<uint64_t>(1024, n / 12)));   // (round 6: the floor was 2048 -- with the pool kernel at 59 us per 4096 sentences a 4096-sentence call runs 182 -> 174 us as four chunks)
    const bool pinned_in = (offsets[n] - offsets[0]) != 0 && is_pinned_host(utf8) && is_pinned_host(offsets);
    constexpr int MAX_DEPTH = 16;
    const int DEPTH = (int)std::min<uint64_t>(MAX_DEPTH, std::max<uint64_t>(3, hooks.depth));
    PipeJob jobs[MAX_DEPTH];
    std::atomic<int> outstanding{0};
    int rc = KGPU_OK;
    uint64_t done = 0, tok_done = 0;
    bool overflow = false;
    tok_offsets[0] = 0;
    int head = 0, inflight = 0;  // jobs[head .. head + inflight) (mod DEPTH) are active, oldest first
    while (!rc && (done < n || (n == 0 && done == 0 && inflight == 0))) {
        if (inflight == DEPTH - 2) {  // two slots stay out of the GPU pipeline: their blocks are being expanded
            const double t0 = now();
            rc = pipe_finish(jobs[head], utf8, offsets, tokens, token_capacity, tok_offsets, status, tok_done, overflow, outstanding);
            t_finish += now() - t0;
            head = (head + 1) % DEPTH; --inflight;
            if (rc) break;
        }
        PipeJob &j = jobs[(head + inflight) % DEPTH];
        if (!j.c) {
            {
                std::lock_guard<std::mutex> g(d->pool_mu);
                if (!d->pool.empty()) { j.c = d->pool.back(); d->pool.pop_back(); }
            }
            if (!j.c && (rc = kgpu_ctx_create(d, nullptr, &j.c))) break;
        }
        uint64_t m = 0;
        while (done + m < n && m < CHUNK_SENTS && (m == 0 || offsets[done + m + 1] - offsets[done] <= CHUNK_BYTES)) ++m;
        j.lo = done; j.m = m;
        const double t0 = now();
        j.stream = expand_stream_wanted((uint64_t)n * 2);   // by the CALL's size (32 768 tokens ~ 16 384 sentences and more): a 4096-sentence call's records are read back at once
        if ((rc = pipe_submit(j, utf8, offsets, pinned_in))) break;
        t_submit += now() - t0;
        ++inflight;
        done += m;
        if (n == 0) break;
    }
    while (inflight) {  // drain in order (also after an error: the contexts go back to the pool idle)
        int r2 = pipe_finish(jobs[head], utf8, offsets, tokens, token_capacity, tok_offsets, status, tok_done, overflow, outstanding);
        if (!rc) rc = r2;
        head = (head + 1) % DEPTH; --inflight;
    }
    const double t_drained = now();
    workers().wait_zero(outstanding);
    t_tail = now() - t_drained;
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        for (int k = 0; k < DEPTH; ++k)
            if (jobs[k].c) d->pool.push_back(jobs[k].c);
    }
    if (trace)
        fprintf(stderr, "kgpu_tokenize_batch: %llu sentences in %.3f ms: submit %.3f (input staging + enqueue), finish while filling %.3f, drain %.3f, "
                        "waiting for the expansion %.3f; chunks of <= %llu sentences, depth %d\n",
                (unsigned long long)n, now() - t_call, t_submit, t_finish, t_drained - t_call - t_submit - t_finish, t_tail, (unsigned long long)CHUNK_SENTS, DEPTH);
    if (n_tokens) *n_tokens = tok_done;
    if (!rc && overflow) {
        set_error("token buffer too small: need %llu, capacity %llu", (unsigned long long)tok_done, (unsigned long long)token_capacity);
        return KGPU_ERR_CAPACITY;
    }
    return rc;
}

// ----------------------------------------------------------------- lattice dump
extern "C" int kgpu_lattice_dump(kgpu_dict *d, const uint8_t *utf8, uint64_t len, kgpu_lattice *out) {
    if (!d || !out || (len && !utf8)) { set_error("kgpu_lattice_dump: null argument"); return KGPU_ERR_INVALID_ARG; }
    if (len >= (1ull << 31)) { set_error("kgpu_lattice_dump: sentence too long"); return KGPU_ERR_INVALID_ARG; }
    *out = kgpu_lattice{};
    HIPCHECK(hipSetDevice(d->device));
    kgpu_ctx *c = nullptr;
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        if (!d->pool.empty()) { c = d->pool.back(); d->pool.pop_back(); }
    }
    int rc = KGPU_OK;
    if (!c && (rc = kgpu_ctx_create(d, nullptr, &c))) return rc;
    auto give_back = [&]() { std::lock_guard<std::mutex> g(d->pool_mu); d->pool.push_back(c); };
    const uint64_t offs[2] = {0, len};
    Control hc{};
    if ((c->pending && (rc = kgpu_ctx_sync(c, nullptr)) != KGPU_OK && rc != KGPU_ERR_CAPACITY) ||
        (rc = c->arena.ensure(ARENA_INITIAL)) || (rc = c->stage.ensure((size_t)(len + 2) * sizeof(kgpu_token) + 64)) ||
        (rc = c->tok_count.ensure(8)) || (rc = c->in_utf8.ensure((size_t)len + 16)) || (rc = c->in_off.ensure(16)) ||
        (rc = c->out_status.ensure(16))) { give_back(); return rc; }
    auto fail = [&](hipError_t e, const char *what) { set_error("kgpu_lattice_dump: %s: %s", what, hipGetErrorString(e)); c->ctl_dirty = true; give_back(); return KGPU_ERR_HIP; };
    hipError_t e;
    if (len && (e = hipMemcpyAsync(c->in_utf8.p, utf8, (size_t)len, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(e, "H2D");
    c->h2d_queued = true;
    if ((e = hipMemcpyAsync(c->in_off.p, offs, 16, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(e, "H2D");
    for (;;) {
        BatchArgs a{};
        a.utf8 = (const uint8_t *)c->in_utf8.p; a.offsets = (const uint64_t *)c->in_off.p; a.n = 1; a.ctl = c->d_ctl;
        a.arena = (uint8_t *)c->arena.p; a.arena_bytes = c->arena.bytes;
        a.stage = (kgpu_token *)c->stage.p; a.tok_count = (uint32_t *)c->tok_count.p; a.status = (uint8_t *)c->out_status.p;
        a.dump_lattice = 1;
        c->ctl_dirty = true;  // no scan kernel behind this launch: the next batch zeroes the block itself
        if ((e = hipMemsetAsync(c->d_ctl, 0, sizeof(Control), c->stream)) != hipSuccess) return fail(e, "memset");
        if ((e = (hipError_t)launch_general_only(d->view, a, c->stream)) != hipSuccess) return fail(e, "launch");
        if ((e = hipMemcpyAsync(&hc, c->d_ctl, sizeof(Control), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return fail(e, "D2H");
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return fail(e, "sync");
        if (!hc.arena_overflow) break;
        size_t want = c->arena.bytes * 2;
        if (want > ARENA_MAX) { set_error("scratch arena exceeded %zu bytes", ARENA_MAX); give_back(); return KGPU_ERR_INTERNAL; }
        if ((rc = c->arena.ensure(want))) { give_back(); return rc; }
    }
    if (!hc.dump[5]) { set_error("kgpu_lattice_dump: the sentence is not valid UTF-8"); give_back(); return KGPU_ERR_INVALID_ARG; }
    const uint64_t B = hc.dump[2], C = hc.dump[3], N = hc.dump[4], na = B + 4;
    std::vector<uint32_t> cbyte(C + 1), boff(C + 2), pre(N);
    std::vector<uint32_t> nodeA(4 * N), bucket(4 * N), nodeB(2 * N);
    const uint8_t *sa = (const uint8_t *)c->arena.p + hc.dump[0], *sn = (const uint8_t *)c->arena.p + hc.dump[1];
    // slab layouts: k_tokenize_general (kgpu_kernels.hip): cbyte | uspan | nb | boff | ... (u32[na] each); nodeA | bucket | nodeB | pre
    if ((e = hipMemcpy(cbyte.data(), sa, (C + 1) * 4, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(boff.data(), sa + 3 * na * 4, (C + 2) * 4, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(nodeA.data(), sn, N * 16, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(bucket.data(), sn + N * 16, N * 16, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(nodeB.data(), sn + N * 32, N * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(pre.data(), sn + N * 40, N * 4, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "slab read-back");
    give_back();
    out->n_nodes = N; out->n_positions = C + 2;
    out->nodes = (kgpu_lattice_node *)calloc((size_t)N, sizeof(kgpu_lattice_node));
    out->edge_offsets = (uint32_t *)calloc((size_t)C + 3, 4);
    out->edge_nodes = (uint32_t *)calloc((size_t)N, 4);
    if (!out->nodes || !out->edge_offsets || !out->edge_nodes) { kgpu_lattice_free(out); set_error("kgpu_lattice_dump: out of memory"); return KGPU_ERR_INTERNAL; }
    auto orig = [](const std::vector<uint32_t> &inv, uint32_t r) { return (int16_t)(inv.empty() || r >= inv.size() ? r : inv[r]); };
    for (uint64_t t = 0; t < N; ++t) {
        kgpu_lattice_node &nd = out->nodes[t];
        if (t == 0) { nd.pre = -1; continue; }  // BOS: Dummy at 0 with Morph(0, 0, 0); dp None
        const uint32_t *A = &nodeA[4 * t];
        const int32_t sid = (int32_t)A[3];
        const uint32_t st = nodeB[2 * t], en = nodeB[2 * t + 1];
        nd.id = sid < 0 ? -sid : sid;
        nd.cls = sid > 0 ? KGPU_CLASS_KNOWN : sid < 0 ? KGPU_CLASS_UNKNOWN : KGPU_CLASS_DUMMY;
        nd.char_pos = st; nd.end_char = en; nd.byte_pos = cbyte[st]; nd.byte_len = cbyte[en] - cbyte[st];
        if (sid != 0) { nd.left_id = orig(d->left_of_rank, A[0] & 0xFFFFu); nd.right_id = orig(d->right_of_rank, A[0] >> 16); nd.cost = (int16_t)(int32_t)A[1]; }
        nd.dp = A[2] == 0xFFFFFFFFu ? (int32_t)hc.dump[6] : (int32_t)bucket[4 * A[2]];
        nd.pre = pre[t] == 0xFFFFFFFFu ? -1 : (int32_t)pre[t];
    }
    // edges[e] = nodes ending at e, ascending node index (the kernel fills a bucket in arrival order)
    for (uint64_t e2 = 0; e2 <= C + 1; ++e2) out->edge_offsets[e2] = e2 <= C + 1 && e2 < boff.size() ? boff[e2] : 0;
    out->edge_offsets[C + 1] = (uint32_t)(N - 1); out->edge_offsets[C + 2] = (uint32_t)N;
    for (uint64_t sl = 0; sl + 1 < N; ++sl) out->edge_nodes[sl] = bucket[4 * sl + 2];
    out->edge_nodes[N - 1] = (uint32_t)(N - 1);
    for (uint64_t e2 = 0; e2 <= C; ++e2) std::sort(out->edge_nodes + out->edge_offsets[e2], out->edge_nodes + out->edge_offsets[e2 + 1]);
    return KGPU_OK;
}

extern "C" void kgpu_lattice_free(kgpu_lattice *l) {
    if (!l) return;
    free(l->nodes); free(l->edge_offsets); free(l->edge_nodes);
    *l = kgpu_lattice{};
}

// Pinned, device-visible host memory for the buffers of kgpu_tokenize_batch: the copies then run as DMA
// at PCIe speed and overlap the kernels (pageable memory is staged by the runtime, synchronously).
extern "C" void *kgpu_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)(bytes ? bytes : 1), hipHostMallocDefault) != hipSuccess) { set_error("kgpu_host_alloc: %llu bytes failed", (unsigned long long)bytes); return nullptr; }
    return p;
}
extern "C" void kgpu_host_free(void *p) { if (p) (void)hipHostFree(p); }

// ---- measurement / test only (bench.py's concurrent_callers leg, tests/test_gpu_concurrent.py; not part of include/kanpyo_gpu.h): `threads` host
// threads call kgpu_tokenize_batch in a loop, thread t with n_pattern[t % n_pat] sentences per call (the reference's shape is 1: src/bin/kanpyo.rs:106-126),
// walking round the corpus from its own starting point.  With `expect_tokens` / `expect_offsets` (the whole corpus tokenized once, e.g. by the oracle)
// every call's records are compared: stats[4] counts the calls that differ.  stats: [0] wall seconds, [1] p50 / [2] p99 / [3] mean call latency in us,
// [4] mismatching calls, [5] calls made, [6] sentences tokenized.  Python threads cannot drive this: the GIL serialises what surrounds each call.
extern "C" int kgpu_debug_concurrent_callers(kgpu_dict *d, const uint8_t *utf8, const uint64_t *offsets, uint64_t n_sentences, int threads, int calls_per_thread,
                                             const int *n_pattern, int n_pat, const kgpu_token *expect_tokens, const uint64_t *expect_offsets, double *stats) {
    if (!d || !offsets || !n_sentences || threads < 1 || threads > 1024 || calls_per_thread < 1 || !n_pattern || n_pat < 1 || !stats) { set_error("kgpu_debug_concurrent_callers: bad argument"); return KGPU_ERR_INVALID_ARG; }
    std::vector<std::vector<float>> lat((size_t)threads);
    std::vector<uint64_t> bad((size_t)threads, 0), sent((size_t)threads, 0), cpu((size_t)threads, 0);   // cpu: the thread's own CPU time over its calls (ns)
    std::vector<int> rcs((size_t)threads, KGPU_OK);
    std::vector<std::string> errs((size_t)threads);
    // The start gate sleeps (a futex word), it does not spin: a hundred threads yielding in a loop while the rest are created burn, each on its own CPU of the
    // host, a good part of a 16-CPU cgroup quota's 100 ms period before the first call is made -- and the period's remainder is then spent throttled.
    std::atomic<int> ready{0};
    std::atomic<uint32_t> go{0};
    auto now_us = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
    auto body = [&](int t) {
        const uint64_t npc = (uint64_t)std::max(1, n_pattern[t % n_pat]);
        uint64_t maxb = 0;
        for (uint64_t i = 0; i < n_sentences; ++i) maxb = std::max(maxb, offsets[i + 1] - offsets[i]);
        std::vector<kgpu_token> tok((size_t)(npc * (maxb + 1) + 8));
        std::vector<uint64_t> toff((size_t)npc + 1), off2((size_t)npc + 1);
        std::vector<uint8_t> st((size_t)npc + 1), text;
        lat[(size_t)t].reserve((size_t)calls_per_thread);
        uint64_t at = ((uint64_t)t * 7919u) % n_sentences;
        ready.fetch_add(1);
        while (go.load(std::memory_order_acquire) == 0) futex_wait(&go, 0);
        const uint64_t c0 = cpu_ns();
        struct CpuOut { uint64_t &out, c0; ~CpuOut() { out = cpu_ns() - c0; } } cpu_out{cpu[(size_t)t], c0};
        for (int k = 0; k < calls_per_thread; ++k) {
            if (at + npc > n_sentences) at = 0;
            const uint64_t m = std::min(npc, n_sentences - at);
            uint64_t got = 0;
            const double t0 = now_us();
            const int rc = kgpu_tokenize_batch(d, utf8, offsets + at, m, tok.data(), tok.size(), toff.data(), st.data(), &got);
            lat[(size_t)t].push_back((float)(now_us() - t0));
            if (rc) { rcs[(size_t)t] = rc; errs[(size_t)t] = kgpu_last_error(); return; }
            sent[(size_t)t] += m;
            if (expect_tokens && expect_offsets) {
                const uint64_t e0 = expect_offsets[at], en = expect_offsets[at + m] - e0;
                bool same = got == en && std::memcmp(tok.data(), expect_tokens + e0, (size_t)en * sizeof(kgpu_token)) == 0;
                for (uint64_t i = 0; same && i <= m; ++i) same = toff[(size_t)i] == expect_offsets[at + i] - e0;
                if (!same) bad[(size_t)t]++;
            }
            at += m;
        }
    };
    std::vector<std::thread> th;
    try { for (int t = 0; t < threads; ++t) th.emplace_back(body, t); }
    catch (...) { go.store(2); futex_wake_all(&go); for (auto &x : th) x.join(); set_error("kgpu_debug_concurrent_callers: could not start %d threads", threads); return KGPU_ERR_INTERNAL; }
    while (ready.load() < threads) short_sleep_us(50);
    const double t0 = now_us();
    go.store(1, std::memory_order_release);
    futex_wake_all(&go);
    for (auto &x : th) x.join();
    const double wall = (now_us() - t0) * 1e-6;
    for (int t = 0; t < threads; ++t) if (rcs[(size_t)t]) { set_error("%s", errs[(size_t)t].c_str()); return rcs[(size_t)t]; }
    std::vector<float> all;
    uint64_t nbad = 0, nsent = 0;
    for (int t = 0; t < threads; ++t) { all.insert(all.end(), lat[(size_t)t].begin(), lat[(size_t)t].end()); nbad += bad[(size_t)t]; nsent += sent[(size_t)t]; }
    std::sort(all.begin(), all.end());
    double mean = 0; for (float x : all) mean += x;
    stats[0] = wall; stats[1] = all[all.size() / 2]; stats[2] = all[(size_t)((double)all.size() * 0.99)]; stats[3] = mean / (double)all.size();
    stats[4] = (double)nbad; stats[5] = (double)all.size(); stats[6] = (double)nsent;
    { uint64_t c = 0; for (uint64_t x : cpu) c += x; stats[7] = (double)c * 1e-9; }   // CPU seconds of the calling threads, start gate and thread start-up left out
    return KGPU_OK;
}

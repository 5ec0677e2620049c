// kanpyo_amd/csrc/kgpu_multi.cpp -- the multi-device entry points of include/kanpyo_gpu.h.
//
// Sentences are independent (reference src/tokenizer.rs:16: &self, one lattice per call), so a node's GPUs share nothing but the
// dictionary, which is replicated: sentence i goes to device i mod G (BASELINE cfg 4: "sharded round-robin"), and the only
// cross-device step is putting the results back in the caller's order.  No collective on the data path.
//
//   kgpu_tokenize_batch_multi   host buffers in, dense 24-byte records out, original order -- what a Rust host calls for a corpus.
//                               One host thread per device runs that device's chunk pipeline (gather the shard's sentences into pinned
//                               staging, one H2D copy, the launch chain, 8-byte records written into mapped host memory by the
//                               compaction kernel); the calling thread merges: per super-chunk of G x M sentences it turns the G local
//                               offset tables into the global one and hands the expansion into the caller's buffer to the worker threads.
//   kgpu_multi_*                device-resident shards in, 8-byte records gathered on the root device: every device's compaction kernel
//                               stores its records straight into the root's memory over xGMI (peer access) -- the stores are the gather.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kgpu_runtime.h"

namespace kgpu {

// ---- the merge of G shards' results into the caller's order (host only: no device call in here; tests/test_multi_merge_cpu.py drives it without a GPU).
// Global sentence j of a super-chunk lives in shard j mod G at local index j / G.  Per slice of the super-chunk: its token total is a difference of two
// entries of every shard's offset table (O(G), on the merging thread), an exclusive scan over the slices gives every slice its base, and one task per
// slice then walks the G shard cursors once -- offsets, expansion of the 8-byte records and status bytes in the same pass, no division per sentence.
struct MergeSrc { const kgpu_token8 *rec; const uint32_t *first; const uint64_t *toff; const uint8_t *st; };

// sentences j < x of a super-chunk with j mod G == g
static inline uint64_t shard_prefix(uint64_t x, uint64_t g, uint64_t G) { return (x + G - 1 - g) / G; }

// slice_base[t] = tokens of the super-chunk in front of slice t (t = 0 .. nt; [nt] = the super-chunk's total)
void merge_plan(const MergeSrc *src, int G, uint64_t cnt, uint64_t slice, std::vector<uint64_t> &slice_base) {
    const uint64_t nt = (cnt + slice - 1) / slice, Gu = (uint64_t)G;
    slice_base.assign((size_t)nt + 1, 0);
    uint64_t run = 0;
    for (uint64_t t = 0; t < nt; ++t) {
        slice_base[(size_t)t] = run;
        const uint64_t a = t * slice, b = std::min(cnt, a + slice);
        for (uint64_t g = 0; g < Gu; ++g) run += src[g].toff[shard_prefix(b, g, Gu)] - src[g].toff[shard_prefix(a, g, Gu)];
    }
    slice_base[(size_t)nt] = run;
}

// sentences [a, b) of the super-chunk: tok_offsets[j] (already pointing at the super-chunk's first entry), the records, status bytes.  base = tokens in
// front of sentence a, in the caller's numbering.  Records: 24-byte kgpu_token into `tokens` (the caller's whole array), or -- the compact form,
// kgpu_tokenize_batch_multi_compact -- the shards' 8-byte records as they are into `out8` plus every sentence's (position, start) of its first token into
// `first_out` (pointing at the super-chunk's first sentence, like tok_offsets); both null: offsets and status only.
void merge_slice(const MergeSrc *src, int G, uint64_t a, uint64_t b, uint64_t base, kgpu_token *tokens, uint64_t *tok_offsets, uint8_t *status, bool stream,
                 kgpu_token8 *out8 = nullptr, uint32_t *first_out = nullptr) {
    const uint64_t Gu = (uint64_t)G;
    uint64_t g = a % Gu, k = a / Gu, run = base;
    for (uint64_t j = a; j < b; ++j) {
        const MergeSrc &sx = src[g];
        const uint64_t t0 = sx.toff[k], t1 = sx.toff[k + 1];
        tok_offsets[j] = run;
        if (tokens) expand_tokens(sx.rec + t0, sx.toff + k, sx.first + 2 * k, 1, tokens + run, stream);   // (one sentence: the records of shard g's local sentence k)
        if (out8) {
#if defined(__x86_64__)
            if (stream) {   // (as the 24-byte expansion: hundreds of megabytes nobody reads back soon -- an ordinary store first reads every line it overwrites)
                const long long *in = (const long long *)(sx.rec + t0);
                long long *o = (long long *)(out8 + run);
                for (uint64_t k = 0, e = t1 - t0; k < e; ++k) _mm_stream_si64(o + k, in[k]);
            } else
#endif
            if (t1 > t0) std::memcpy(out8 + run, sx.rec + t0, (size_t)(t1 - t0) * sizeof(kgpu_token8));
            first_out[2 * j] = sx.first[2 * k]; first_out[2 * j + 1] = sx.first[2 * k + 1];
        }
        if (status) status[j] = sx.st[k];
        run += t1 - t0;
        if (++g == Gu) { g = 0; ++k; }
    }
    if (stream && (tokens || out8)) expand_fence();
}

// The caller's current device, restored on every exit path: the multi-device entry points visit every device on the calling thread.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

}  // namespace kgpu

namespace {

constexpr int NSLOT = 8;       // chunks of one device between submission and the end of their expansion
constexpr int GPU_DEPTH = 6;   // ... of which in the device pipeline

struct ShardJob {
    kgpu_ctx *c = nullptr;
    uint64_t m = 0, cap = 0;
    size_t off_first = 0, off_toff = 0, off_status = 0;  // inside pin_out: records | first | token offsets | status
};

struct MultiCall {
    int G = 0;
    kgpu_dict *const *dicts = nullptr;
    const uint8_t *utf8 = nullptr; const uint64_t *offsets = nullptr; uint64_t n = 0;
    uint64_t S = 0, NC = 0;                       // sentences per super-chunk (a multiple of G), super-chunks
    std::vector<ShardJob> jobs;                   // [g * NSLOT + slot]
    std::vector<std::atomic<int>> ready;          // [chunk % NSLOT]: devices through with that chunk
    std::atomic<int> slot_tasks[NSLOT];           // expansion tasks still reading the slot's result blocks
    std::atomic<int64_t> merged{0};               // super-chunks merged (their expansion tasks submitted)
    std::atomic<int> failed{0};                   // first error code
    char err[400] = "";
    std::mutex mu; std::condition_variable cv;    // ready / merged / failed changes
    MultiCall() : ready(NSLOT) { for (auto &r : ready) r.store(0); for (auto &t : slot_tasks) t.store(0); }

    void fail(int rc) {
        int zero = 0;
        if (failed.compare_exchange_strong(zero, rc)) snprintf(err, sizeof err, "%s", kgpu_last_error());
        std::lock_guard<std::mutex> g(mu);
        cv.notify_all();
    }
    uint64_t chunk_lo(uint64_t c) const { return c * S; }
    uint64_t chunk_hi(uint64_t c) const { return std::min(n, (c + 1) * S); }
    uint64_t shard_count(uint64_t c, int g) const {  // sentences of super-chunk c that are device g's
        const uint64_t cnt = chunk_hi(c) - chunk_lo(c);
        return cnt > (uint64_t)g ? (cnt - (uint64_t)g + (uint64_t)G - 1) / (uint64_t)G : 0;
    }
};

// Device g's share of super-chunk c: gather its sentences (every G-th) into the context's pinned staging block as [offsets | bytes], one H2D copy, the launch chain.
int shard_submit(MultiCall &mc, int g, uint64_t c) {
    ShardJob &j = mc.jobs[(size_t)g * NSLOT + (size_t)(c % NSLOT)];
    kgpu_ctx *ctx = j.c;
    const uint64_t lo = mc.chunk_lo(c), m = mc.shard_count(c, g);
    j.m = m;
    uint64_t total = 0;
    for (uint64_t k = 0; k < m; ++k) { const uint64_t i = lo + (uint64_t)g + k * (uint64_t)mc.G; total += mc.offsets[i + 1] - mc.offsets[i]; }
    j.cap = total + m + 1;  // tokens <= chars + 1 <= bytes + 1 per sentence: never too small
    const size_t in_off_bytes = ((size_t)(m + 1) * 8 + 63) & ~(size_t)63, in_bytes = in_off_bytes + (size_t)total + 16;
    j.off_first = ((size_t)j.cap * 8 + 63) & ~(size_t)63;
    j.off_toff = j.off_first + (((size_t)m * 8 + 63) & ~(size_t)63);
    j.off_status = j.off_toff + (((size_t)(m + 1) * 8 + 63) & ~(size_t)63);
    int rc;
    if ((rc = ctx->in_block.ensure(in_bytes)) || (rc = ctx->pin_out.ensure(j.off_status + (size_t)m + 64, true)) || (rc = ctx->out_status.ensure((size_t)m + 16)) ||
        (rc = ctx->out_off.ensure((size_t)(m + 1) * 8)) || (rc = ctx->pin_in.ensure(in_bytes, false)))
        return rc;
    uint64_t *h_off = (uint64_t *)ctx->pin_in.h;
    uint8_t *h_txt = (uint8_t *)ctx->pin_in.h + in_off_bytes;
    {
        uint64_t at = 0;
        for (uint64_t k = 0; k < m; ++k) { const uint64_t i = lo + (uint64_t)g + k * (uint64_t)mc.G; h_off[k] = at; at += mc.offsets[i + 1] - mc.offsets[i]; }
        h_off[m] = at;
    }
    {   // the bytes, in a few slices with the workers' help (one thread gathers ~40 M sentences/s: less than its device tokenizes)
        const uint64_t PIECE = 2048;
        const int np = (int)std::min<uint64_t>(4, (m + PIECE - 1) / PIECE);
        std::atomic<int> left{std::max(np - 1, 0)};
        auto piece = [&mc, lo, g, h_off, h_txt](uint64_t k0, uint64_t k1) {
            for (uint64_t k = k0; k < k1; ++k) {
                const uint64_t i = lo + (uint64_t)g + k * (uint64_t)mc.G, len = mc.offsets[i + 1] - mc.offsets[i];
                if (len) std::memcpy(h_txt + h_off[k], mc.utf8 + mc.offsets[i], (size_t)len);
            }
        };
        const uint64_t each = np > 0 ? (m + (uint64_t)np - 1) / (uint64_t)np : m;
        for (int p = 1; p < np; ++p) {
            std::atomic<int> *l = &left;
            const uint64_t k0 = (uint64_t)p * each, k1 = std::min(m, k0 + each);
            workers().submit([=] { piece(k0, k1); workers().task_done(*l); });
        }
        piece(0, std::min(m, each));
        if (np > 1) workers().wait_zero(left);
    }
    hipError_t e;
    uint8_t *dblk = (uint8_t *)ctx->in_block.p;
    ctx->h2d_queued = true;
    if ((e = hipMemcpyAsync(dblk, ctx->pin_in.h, in_off_bytes + (size_t)total, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) { set_error("H2D input block: %s", hipGetErrorString(e)); return KGPU_ERR_HIP; }
    uint8_t *po = (uint8_t *)ctx->pin_out.d;
    return tokenize_device_impl(ctx, dblk + in_off_bytes, (const uint64_t *)dblk, m, total, nullptr, (kgpu_token8 *)po, (uint32_t *)(po + j.off_first), po + j.off_status,
                                (uint64_t *)(po + j.off_toff), j.cap, (uint64_t *)ctx->out_off.p, (uint8_t *)ctx->out_status.p, "kgpu_tokenize_batch_multi");
}

void device_thread(MultiCall *pmc, int g) {
    MultiCall &mc = *pmc;
    if (hipSetDevice(mc.dicts[g]->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", mc.dicts[g]->device); mc.fail(KGPU_ERR_HIP); return; }
    auto finish = [&](uint64_t c) {   // the chunk's records are in host memory when its context is through
        ShardJob &j = mc.jobs[(size_t)g * NSLOT + (size_t)(c % NSLOT)];
        uint64_t got = 0;
        int rc = kgpu_ctx_sync(j.c, &got);
        if (rc == KGPU_ERR_CAPACITY && j.c->h_ctl->pack_overflow) { set_error("kgpu_tokenize_batch_multi: a token does not fit the 8-byte record (more than 4095 chars or 262143 bytes); tokenize this input per device with kgpu_tokenize_batch"); rc = KGPU_ERR_INTERNAL; }
        if (rc) { mc.fail(rc); return false; }
        mc.ready[c % NSLOT].fetch_add(1, std::memory_order_acq_rel);
        { std::lock_guard<std::mutex> l(mc.mu); }
        mc.cv.notify_all();
        return true;
    };
    uint64_t done = 0;  // chunks finished
    for (uint64_t c = 0; c < mc.NC && !mc.failed.load(std::memory_order_acquire); ++c) {
        if (c >= (uint64_t)GPU_DEPTH) { if (!finish(done)) return; ++done; }
        if (c >= (uint64_t)NSLOT) {   // the slot's previous chunk must have been merged and expanded
            std::unique_lock<std::mutex> l(mc.mu);
            mc.cv.wait(l, [&] { return mc.failed.load() || mc.merged.load(std::memory_order_acquire) > (int64_t)(c - NSLOT); });
            l.unlock();
            if (mc.failed.load()) return;
            workers().wait_zero(mc.slot_tasks[c % NSLOT]);
        }
        const int rc = shard_submit(mc, g, c);
        if (rc) { mc.fail(rc); break; }
    }
    // (after a failure the contexts still hold pending batches: kgpu_ctx_sync below retires them so that they go back to the pool idle)
    while (done < mc.NC) {
        if (mc.failed.load()) { for (int s = 0; s < NSLOT; ++s) { kgpu_ctx *c = mc.jobs[(size_t)g * NSLOT + s].c; if (c && c->pending) (void)kgpu_ctx_sync(c, nullptr); } return; }
        if (!finish(done)) return;
        ++done;
    }
}

}  // namespace

// Both host-buffer forms: tokens (24-byte records) or tokens8 + first (the compact form), never both.
static int multi_impl(kgpu_dict *const *dicts, int n_dicts, const uint8_t *utf8, const uint64_t *offsets, uint64_t n, kgpu_token *tokens, kgpu_token8 *tokens8,
                      uint32_t *first, uint64_t token_capacity, uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens) {
    const bool compact = tokens8 != nullptr || first != nullptr;
    if (!dicts || n_dicts < 1 || n_dicts > 64 || !offsets || !tok_offsets || (token_capacity && !tokens && !tokens8) || (compact && n && !first)) {
        set_error("kgpu_tokenize_batch_multi: bad argument");
        return KGPU_ERR_INVALID_ARG;
    }
    for (int g = 0; g < n_dicts; ++g) if (!dicts[g]) { set_error("kgpu_tokenize_batch_multi: null dictionary handle %d", g); return KGPU_ERR_INVALID_ARG; }
    DeviceGuard keep_callers_device;   // (above every path that touches a device: the one-dictionary shortcut included)
    if (n_dicts == 1 && !compact) return kgpu_tokenize_batch(dicts[0], utf8, offsets, n, tokens, token_capacity, tok_offsets, status, n_tokens);
    for (uint64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) { set_error("kgpu_tokenize_batch_multi: offsets not monotone at %llu", (unsigned long long)i); return KGPU_ERR_INVALID_ARG; }
    if (offsets[n] - offsets[0] && !utf8) { set_error("kgpu_tokenize_batch_multi: null utf8"); return KGPU_ERR_INVALID_ARG; }
    tok_offsets[0] = 0;
    if (n_tokens) *n_tokens = 0;
    if (n == 0) return KGPU_OK;
    if (workers().start() == 0) { set_error("kgpu_tokenize_batch_multi: no worker threads could be started"); return KGPU_ERR_INTERNAL; }

    const int G = n_dicts;
    MultiCall mc;
    mc.G = G; mc.dicts = dicts; mc.utf8 = utf8; mc.offsets = offsets; mc.n = n;
    {   // per-device chunks of <= 8192 sentences and <= 2 MB (the single-device pipeline's sizes); a super-chunk is G of them
        const uint64_t m_env = test_hooks().multi_chunk_sents;  // (tests: small chunks -> many super-chunks)
        uint64_t M = m_env ? m_env : std::min<uint64_t>(8192, std::max<uint64_t>(2048, n / (12 * (uint64_t)G)));
        const uint64_t avg = (offsets[n] - offsets[0]) / n + 1;
        M = std::max<uint64_t>(1, std::min<uint64_t>(M, (2ull << 20) / avg));
        mc.S = M * (uint64_t)G;
        mc.NC = (n + mc.S - 1) / mc.S;
    }
    mc.jobs.resize((size_t)G * NSLOT);
    int rc = KGPU_OK;
    for (int g = 0; g < G && !rc; ++g) {
        if (hipSetDevice(dicts[g]->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", dicts[g]->device); rc = KGPU_ERR_HIP; break; }
        for (int s = 0; s < NSLOT && (uint64_t)s < mc.NC && !rc; ++s) rc = pool_get(dicts[g], &mc.jobs[(size_t)g * NSLOT + s].c);
    }
    std::vector<std::thread> threads;
    if (!rc) {
        try { for (int g = 0; g < G; ++g) threads.emplace_back(device_thread, &mc, g); }
        catch (...) { set_error("kgpu_tokenize_batch_multi: could not start a device thread"); mc.fail(KGPU_ERR_INTERNAL); }
    }

    // ---- merge, in chunk order: local token offsets of the G shards -> the global table, expansion into the caller's records by the workers
    uint64_t tok_done = 0;
    bool overflow = false;
    std::atomic<int> outstanding{0};
    for (uint64_t c = 0; !rc && c < mc.NC && threads.size() == (size_t)G; ++c) {
        {
            std::unique_lock<std::mutex> l(mc.mu);
            mc.cv.wait(l, [&] { return mc.failed.load() || mc.ready[c % NSLOT].load(std::memory_order_acquire) == G; });
        }
        if (mc.failed.load()) break;
        mc.ready[c % NSLOT].store(0, std::memory_order_release);
        const uint64_t lo = mc.chunk_lo(c), cnt = mc.chunk_hi(c) - lo;
        const int slot = (int)(c % NSLOT);
        std::vector<MergeSrc> src((size_t)G);
        for (int g = 0; g < G; ++g) {
            const ShardJob &j = mc.jobs[(size_t)g * NSLOT + slot];
            const uint8_t *ph = (const uint8_t *)j.c->pin_out.h;
            src[(size_t)g] = MergeSrc{(const kgpu_token8 *)ph, (const uint32_t *)(ph + j.off_first), (const uint64_t *)(ph + j.off_toff), ph + j.off_status};
        }
        // slice totals from the shards' offset tables (O(slices x G) on this thread), then one task per slice does everything else
        const uint64_t SLICE = 2048;
        std::vector<uint64_t> slice_base;
        merge_plan(src.data(), G, cnt, SLICE, slice_base);
        const int nt = (int)(slice_base.size() - 1);
        const uint64_t chunk_tokens = slice_base[(size_t)nt];
        if (tok_done + chunk_tokens > token_capacity) overflow = true;   // (from here on: totals and status bytes only)
        const uint64_t base0 = tok_done;
        tok_done += chunk_tokens;
        if (!overflow) tok_offsets[lo + cnt] = tok_done;
        mc.slot_tasks[slot].store(nt, std::memory_order_release);
        outstanding.fetch_add(nt, std::memory_order_acq_rel);
        const bool ovf = overflow;
        for (int t = 0; t < nt; ++t) {
            const uint64_t a = (uint64_t)t * SLICE, b = std::min(cnt, a + SLICE), base = base0 + slice_base[(size_t)t];
            std::atomic<int> *st_ = &mc.slot_tasks[slot], *out_ = &outstanding;
            workers().submit([=] {
                if (!ovf) merge_slice(src.data(), G, a, b, base, tokens, tok_offsets + lo, status ? status + lo : nullptr, expand_stream_wanted(chunk_tokens), tokens8, first ? first + 2 * lo : nullptr);
                else if (status) { uint64_t g = a % (uint64_t)G, k = a / (uint64_t)G; for (uint64_t jx = a; jx < b; ++jx) { status[lo + jx] = src[(size_t)g].st[k]; if (++g == (uint64_t)G) { g = 0; ++k; } } }
                workers().task_done(*st_);
                workers().task_done(*out_);
            });
        }
        mc.merged.store((int64_t)c + 1, std::memory_order_release);
        { std::lock_guard<std::mutex> l(mc.mu); }
        mc.cv.notify_all();
    }
    if (rc) mc.fail(rc);
    for (auto &t : threads) t.join();
    workers().wait_zero(outstanding);
    for (int g = 0; g < G; ++g) {
        (void)hipSetDevice(dicts[g]->device);
        for (int s = 0; s < NSLOT; ++s) {
            kgpu_ctx *c = mc.jobs[(size_t)g * NSLOT + s].c;
            if (!c) continue;
            if (c->pending) (void)kgpu_ctx_sync(c, nullptr);
            pool_put(dicts[g], c);
        }
    }
    if (mc.failed.load()) { set_error("%s", mc.err); return mc.failed.load(); }
    if (n_tokens) *n_tokens = tok_done;
    if (overflow) { set_error("token buffer too small: need %llu, capacity %llu", (unsigned long long)tok_done, (unsigned long long)token_capacity); return KGPU_ERR_CAPACITY; }
    return KGPU_OK;
}

extern "C" int kgpu_tokenize_batch_multi(kgpu_dict *const *dicts, int n_dicts, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                                         kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens) {
    if (token_capacity && !tokens) { set_error("kgpu_tokenize_batch_multi: bad argument"); return KGPU_ERR_INVALID_ARG; }
    return multi_impl(dicts, n_dicts, utf8, offsets, n, tokens, nullptr, nullptr, token_capacity, tok_offsets, status, n_tokens);
}

// The same call with the records left as the devices produce them: 8-byte kgpu_token8 + the first token's (position, start) per sentence, in the caller's
// order.  The 24-byte expansion (kgpu_expand_tokens) writes ~1 KB of host memory per cfg 2 sentence and binds the 24-byte form at about two devices' worth
// of records on 16 CPUs; here the merge moves a third of that and the expansion is the caller's choice (per consumer thread, per sentence, or never).
extern "C" int kgpu_tokenize_batch_multi_compact(kgpu_dict *const *dicts, int n_dicts, const uint8_t *utf8, const uint64_t *offsets, uint64_t n,
                                                 kgpu_token8 *tokens8, uint64_t token_capacity, uint32_t *first, uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens) {
    if (!first && n) { set_error("kgpu_tokenize_batch_multi_compact: bad argument"); return KGPU_ERR_INVALID_ARG; }
    if (token_capacity && !tokens8) { set_error("kgpu_tokenize_batch_multi_compact: bad argument"); return KGPU_ERR_INVALID_ARG; }
    static uint32_t none[2];
    return multi_impl(dicts, n_dicts, utf8, offsets, n, nullptr, tokens8, first ? first : none, token_capacity, tok_offsets, status, n_tokens);
}

// ------------------------------------------------------------------------------------------------ device-resident form
struct kgpu_multi {
    std::vector<kgpu_dict *> dicts;
    int slots = 0;
    std::vector<kgpu_ctx *> ctx;  // [slot * G + g]
};

extern "C" int kgpu_multi_create(kgpu_dict *const *dicts, int n_dicts, int slots, kgpu_multi **out) {
    if (!dicts || !out || n_dicts < 1 || n_dicts > 64 || slots < 1 || slots > 64) { set_error("kgpu_multi_create: bad argument"); return KGPU_ERR_INVALID_ARG; }
    *out = nullptr;
    for (int g = 0; g < n_dicts; ++g) if (!dicts[g]) { set_error("kgpu_multi_create: null dictionary handle %d", g); return KGPU_ERR_INVALID_ARG; }
    DeviceGuard keep_callers_device;
    kgpu_multi *m = new kgpu_multi();
    m->dicts.assign(dicts, dicts + n_dicts);
    m->slots = slots;
    const int root = dicts[0]->device;
    for (int g = 0; g < n_dicts; ++g) {   // the shards' compaction kernels store into the root's memory: peer access from every other device
        if (dicts[g]->device == root) continue;
        if (hipSetDevice(dicts[g]->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", dicts[g]->device); delete m; return KGPU_ERR_HIP; }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dicts[g]->device, root) != hipSuccess || !can) { set_error("kgpu_multi_create: device %d cannot access device %d's memory", dicts[g]->device, root); delete m; return KGPU_ERR_HIP; }
        const hipError_t e = hipDeviceEnablePeerAccess(root, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", dicts[g]->device, root, hipGetErrorString(e)); delete m; return KGPU_ERR_HIP; }
        (void)hipGetLastError();
    }
    m->ctx.assign((size_t)slots * n_dicts, nullptr);
    for (int s = 0; s < slots; ++s)
        for (int g = 0; g < n_dicts; ++g) {
            const int rc = kgpu_ctx_create(dicts[g], nullptr, &m->ctx[(size_t)s * n_dicts + g]);
            if (rc) { kgpu_multi_destroy(m); return rc; }
        }
    *out = m;
    return KGPU_OK;
}

extern "C" void kgpu_multi_destroy(kgpu_multi *m) {
    if (!m) return;
    DeviceGuard keep_callers_device;
    for (kgpu_ctx *c : m->ctx) if (c) kgpu_ctx_destroy(c);
    delete m;
}

extern "C" int kgpu_multi_tokenize_device(kgpu_multi *m, int slot, const uint8_t *const *d_utf8, const uint64_t *const *d_offsets, const uint64_t *n,
                                          const uint64_t *total_bytes, kgpu_token8 *const *root_tokens8, const uint64_t *token_capacity,
                                          uint32_t *const *root_first, uint64_t *const *root_tok_offsets, uint8_t *const *root_status) {
    if (!m || slot < 0 || slot >= m->slots || !d_utf8 || !d_offsets || !n || !total_bytes || !root_tokens8 || !token_capacity || !root_first || !root_tok_offsets || !root_status) {
        set_error("kgpu_multi_tokenize_device: bad argument");
        return KGPU_ERR_INVALID_ARG;
    }
    const int G = (int)m->dicts.size();
    DeviceGuard keep_callers_device;
    for (int g = 0; g < G; ++g) {
        kgpu_ctx *c = m->ctx[(size_t)slot * G + g];
        HIPCHECK(hipSetDevice(m->dicts[g]->device));
        int rc;
        if (c->pending && (rc = kgpu_ctx_sync(c, nullptr)) != KGPU_OK && rc != KGPU_ERR_CAPACITY) return rc;
        if ((rc = c->out_off.ensure((size_t)(n[g] + 1) * 8)) || (rc = c->out_status.ensure((size_t)n[g] + 16))) return rc;
        // token offsets and status bytes are produced in the shard's own HBM (the compaction kernel reads the offsets back) and mirrored to the root
        if ((rc = tokenize_device_impl(c, d_utf8[g], d_offsets[g], n[g], total_bytes[g], nullptr, root_tokens8[g], root_first[g], root_status[g], root_tok_offsets[g],
                                       token_capacity[g], (uint64_t *)c->out_off.p, (uint8_t *)c->out_status.p, "kgpu_multi_tokenize_device")))
            return rc;
    }
    return KGPU_OK;
}

extern "C" int kgpu_multi_sync(kgpu_multi *m, int slot, uint64_t *n_tokens) {
    if (!m || slot < 0 || slot >= m->slots) { set_error("kgpu_multi_sync: bad argument"); return KGPU_ERR_INVALID_ARG; }
    const int G = (int)m->dicts.size();
    DeviceGuard keep_callers_device;
    int first_rc = KGPU_OK;
    for (int g = 0; g < G; ++g) {
        uint64_t got = 0;
        const int rc = kgpu_ctx_sync(m->ctx[(size_t)slot * G + g], &got);
        if (n_tokens) n_tokens[g] = got;
        if (rc && !first_rc) first_rc = rc;
    }
    return first_rc;
}

// ------------------------------------------------------------------------------------------------ test / measurement hook (no device)
// The merge of kgpu_tokenize_batch_multi over ONE super-chunk given as G shard blocks in host memory (what the shards' compaction kernels leave in
// their mapped result blocks), through the same plan + worker-pool tasks: tests/test_multi_merge_cpu.py checks it against a plain loop and times it
// (reps > 1: the same merge repeated, seconds = wall time of all repetitions).  Needs no GPU.
static int debug_merge(int G, uint64_t cnt, const kgpu_token8 *const *rec, const uint32_t *const *first, const uint64_t *const *toff,
                       const uint8_t *const *st, uint64_t slice, int reps, kgpu_token *tokens, kgpu_token8 *tokens8, uint32_t *first_out, uint64_t token_capacity,
                       uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens, double *seconds) {
    if (G < 1 || G > 64 || !rec || !first || !toff || !st || !tok_offsets || slice == 0 || reps < 1 || (tokens8 && !first_out)) { set_error("kgpu_debug_merge_shards: bad argument"); return KGPU_ERR_INVALID_ARG; }
    if (workers().start() == 0) { set_error("kgpu_debug_merge_shards: no worker threads"); return KGPU_ERR_INTERNAL; }
    std::vector<MergeSrc> src((size_t)G);
    for (int g = 0; g < G; ++g) src[(size_t)g] = MergeSrc{rec[g], first[g], toff[g], st[g]};
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t total = 0;
    for (int r = 0; r < reps; ++r) {
        std::vector<uint64_t> slice_base;
        merge_plan(src.data(), G, cnt, slice, slice_base);
        const int nt = (int)(slice_base.size() - 1);
        total = slice_base[(size_t)nt];
        const bool fits = (tokens || tokens8) && total <= token_capacity;
        const bool stream = expand_stream_wanted(total);
        tok_offsets[cnt] = total;
        std::atomic<int> left{nt};
        const MergeSrc *sp = src.data();
        for (int t = 0; t < nt; ++t) {
            const uint64_t a = (uint64_t)t * slice, b = std::min(cnt, a + slice), base = slice_base[(size_t)t];
            std::atomic<int> *l = &left;
            workers().submit([=] { merge_slice(sp, G, a, b, base, fits ? tokens : nullptr, tok_offsets, status, stream, fits ? tokens8 : nullptr, first_out); workers().task_done(*l); });
        }
        workers().wait_zero(left);
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (n_tokens) *n_tokens = total;
    if ((tokens || tokens8) && total > token_capacity) { set_error("token buffer too small: need %llu, capacity %llu", (unsigned long long)total, (unsigned long long)token_capacity); return KGPU_ERR_CAPACITY; }
    return KGPU_OK;
}
extern "C" int kgpu_debug_merge_shards(int G, uint64_t cnt, const kgpu_token8 *const *rec, const uint32_t *const *first, const uint64_t *const *toff,
                                       const uint8_t *const *st, uint64_t slice, int reps, kgpu_token *tokens, uint64_t token_capacity, uint64_t *tok_offsets,
                                       uint8_t *status, uint64_t *n_tokens, double *seconds) {
    return debug_merge(G, cnt, rec, first, toff, st, slice, reps, tokens, nullptr, nullptr, token_capacity, tok_offsets, status, n_tokens, seconds);
}
// ... with the compact form's output (kgpu_tokenize_batch_multi_compact's merge)
extern "C" int kgpu_debug_merge_shards_compact(int G, uint64_t cnt, const kgpu_token8 *const *rec, const uint32_t *const *first, const uint64_t *const *toff,
                                               const uint8_t *const *st, uint64_t slice, int reps, kgpu_token8 *tokens8, uint32_t *first_out, uint64_t token_capacity,
                                               uint64_t *tok_offsets, uint8_t *status, uint64_t *n_tokens, double *seconds) {
    if (!first_out) { set_error("kgpu_debug_merge_shards_compact: bad argument"); return KGPU_ERR_INVALID_ARG; }
    return debug_merge(G, cnt, rec, first, toff, st, slice, reps, nullptr, tokens8, first_out, token_capacity, tok_offsets, status, n_tokens, seconds);
}

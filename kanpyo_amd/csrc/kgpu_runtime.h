// kanpyo_amd/csrc/kgpu_runtime.h -- the host runtime's own types, shared by kgpu_api.cpp (dictionary, contexts, the host-buffer
// entry points) and kgpu_multi.cpp (the multi-device entry points).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "kgpu_internal.h"

#define HIPCHECK(expr)                                                                  \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return KGPU_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

namespace kgpu {

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return KGPU_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4 + 256;
        HIPCHECK(hipMalloc(&p, want));
        bytes = want;
        return KGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// Pinned host memory, optionally device-mapped (the device then reads / writes it over PCIe by itself).
struct PinBuf {
    void *h = nullptr, *d = nullptr; size_t bytes = 0;
    int ensure(size_t need, bool mapped) {
        if (need <= bytes) return KGPU_OK;
        release();
        size_t want = need + need / 4 + 4096;
        if (hipHostMalloc(&h, want, mapped ? hipHostMallocMapped : hipHostMallocDefault) != hipSuccess) { h = nullptr; set_error("pinned allocation of %zu bytes failed", want); return KGPU_ERR_HIP; }
        d = h;
        if (mapped && hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); h = d = nullptr; set_error("hipHostGetDevicePointer failed"); return KGPU_ERR_HIP; }
        bytes = want;
        return KGPU_OK;
    }
    void release() { if (h) (void)hipHostFree(h); h = d = nullptr; bytes = 0; }
};

}  // namespace kgpu

using namespace kgpu;  // (a private header of two translation units that both speak this namespace's vocabulary)

struct Combiner;  // kgpu_api.cpp: concurrent small calls sharing a launch

struct kgpu_dict {
    int device = 0;
    Combiner *combiner = nullptr;
    int combiner_callers() const;  // threads inside the small-call entry point right now
    DictView view{};
    kgpu_dict_info info{};
    std::vector<void *> allocs;
    std::mutex pool_mu;
    std::vector<kgpu_ctx *> pool;
    // LDS bytes reserved per input byte (x256) by the pool kernel before the lattice is known; a
    // property of the dictionary + the text, so it is learnt once and shared by all contexts
    std::atomic<uint32_t> est_q8{64 * 256};   // (round 6: the lattice takes ~52 bytes of LDS per input byte on IPADIC-shaped text; was 80)
    // Batches left for which the second (whole-CU) pool is launched.  Its workgroups need a CU's
    // entire LDS just to start and find their list empty, which stalls them -- and the launches
    // queued behind -- until both 80 KB pools of that CU have drained; so it is only issued while
    // recent batches actually overflowed the first pool (otherwise those rare sentences take the
    // HBM-scratch kernel).  Performance heuristic only: the chain is complete either way.
    std::atomic<int> big_pool_batches{0};
    // Same for the long-sentence kernel (its workgroups hold 32 KB of LDS each): issued while recent
    // batches still had sentences left after the pools.
    std::atomic<int> window_batches{64};  // the windowed kernel: in the chain while recent batches left the pools sentences (starts armed)
    std::atomic<int> tail_batches{0};     // the general kernel behind it: while recent batches left the windowed kernel (or, without one, the pools) sentences
    // Streams handed round-robin to contexts created without one.  HIP multiplexes streams onto three
    // hardware queues: a 4th stream queues behind the 1st and unbalances them (measured -25 %), so any
    // number of contexts shares three streams; each context waits on its own completion event.
    std::vector<hipStream_t> streams;
    unsigned next_stream = 0;
    // A second set for chains that START with the windowed kernel (batches of long sentences: kgpu_api.cpp, ctx_pick_chain): such a launch holds a thousand
    // single-wavefront workgroups for milliseconds and its slots empty out one by one, so the chip fills only when more of them overlap than the four launches
    // the pool kernel wants -- one stream per context, up to eight, created when the first such batch arrives (round 5: cfg 5 2.97 -> 3.96 Gchar/s).
    std::vector<hipStream_t> long_streams;
    unsigned next_long = 0;
    std::atomic<int> long_sentences_in_flight{0};   // sentences of window-first batches between enqueue and completion (decides the two-wavefront form)
    std::atomic<int> long_peak{0};                  // ... its recent maximum (decays by an eighth per enqueue): a caller that keeps eight batches in flight is not
                                                    // mistaken for a lone one by the batch that happens to be enqueued while the others are being collected
    std::vector<uint32_t> left_of_rank, right_of_rank;  // device (ranked) context id -> the dictionary's own; empty = identity
    // One reference for the handle the caller holds plus one per live context: the tables and the shared
    // streams go when the last one does (a context outliving kgpu_dict_destroy keeps working).
    std::atomic<int> refs{1};
};

struct kgpu_ctx {
    kgpu_dict *dict = nullptr;
    hipStream_t stream = nullptr;       // the stream of the pending / next batch (one of the dictionary's shared streams unless the caller gave one)
    hipStream_t short_stream = nullptr, long_stream = nullptr;   // what `stream` alternates between (library-owned streams only)
    bool own_stream = false;            // the caller's stream: never switched
    bool window_first = false;          // the next batch's chain starts with the windowed kernel (no pool launch in front)
    bool roomy = false;                 // ... its pool workgroups run three wavefronts instead of four (four reservations of the learnt size do not fit a pool)
    bool last_team = false;             // the pending batch's chain started with the two-wavefronts-per-sentence form (one more work list in the chain)
    int counted_long = 0;               // what the pending batch added to kgpu_dict::long_sentences_in_flight
    uint32_t win_share_q8 = 0;          // share of the last pool-first batch's sentences that the pools routed to the windowed kernel (x256): an eighth or more
                                        // and the next batch runs on the context's long stream too (cfg 3: a third of the sentences, three quarters of the characters)
    bool long_share = false;            // ... with hysteresis: entered at win_share_q8 >= 32, left below 16
    bool h2d_queued = false;            // a host-buffer path has queued this batch's H2D copy on `stream` (ctx_pick_chain orders the batch behind it if it switches streams)
    hipEvent_t switch_ev = nullptr;     // orders a batch behind what was queued on the stream the context used before
    hipEvent_t done_ev = nullptr;  // recorded behind the batch's last kernel: contexts may share a stream
    Control *d_ctl = nullptr;
    Control *h_ctl = nullptr;  // pinned + device-mapped: the scan kernel publishes the launch's Control block here
    Control *h_ctl_dev = nullptr;  // device-side address of h_ctl
    bool ctl_dirty = true;     // d_ctl must be zeroed by the host (first launch, or after a failed enqueue)
    uint32_t launch_seq = 0;
    int last_pools = 0;        // pool launches issued for the pending batch
    bool last_window = false;  // ... and whether the windowed kernel was
    bool no_window = false;    // the pending batch is a rerun without the windowed kernel (it had flagged Control::window_fail)
    bool last_tail = true;     // ... whether the last-resort launch closed the chain (left out while no recent batch needed the tail)
    bool tail_pass = false;    // the pending launches are the tail of the batch's chain alone (it had been left out and a sentence needed it)
    Control tail_saved{};      // ... what the first pass had published
    unsigned tail_count = 0;   // ... the length of the work list the tail serves (source of an asynchronous copy: lives here)
    int tail_li = 0;           // ... that list's index
    bool tail_had_window = false;  // ... whether the first pass had the windowed kernel in its chain
    uint32_t event_every = 1;  // KGPU_PROFILE_SAMPLED: HIP events on every 4th launch only
    DevBuf arena, stage, tok_count;
    // host-buffer path staging
    DevBuf in_utf8, in_off, out_tok, out_off, out_status;
    PinBuf pin_in, pin_out;       // large host calls: input staging (offsets | bytes), mapped result block (records | first | token offsets | status)
    DevBuf in_block;              // ... and the device copy of the input block
    // single-launch small calls: one pinned, device-mapped block (input | offsets | tokens | token offsets | status)
    uint8_t *sm_host = nullptr, *sm_dev = nullptr;
    uint32_t sm_seq = 0;
    // last enqueued batch (for the arena-overflow retry and for sync)
    BatchArgs last{};
    bool pending = false;
    LaunchPlan plan{};
    DevBuf ovf;
    DevBuf stat_slots;                             // profiling runs: per-wavefront counters of the pool kernel (BatchArgs::stat_slots)
    std::vector<unsigned long long> stat_host;
    // profiling
    bool profiling = false;   // KGPU_PROFILE_EVENTS
    bool count_work = false;  // KGPU_PROFILE_WORK
    bool count_no_t = false;  // KGPU_PROFILE_NO_T
    uint32_t stop_after = 0;  // kgpu_ctx_set_ablation: measurement mode, 0 = off
    kgpu_work work{};
    uint64_t phase[10] = {0};
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    kgpu_profile prof{};
    kgpu_routing rt{};
};


#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kgpu {

// Host side of the 8-byte records (include/kanpyo_gpu.h, kgpu_token8): n sentences' records -> 24-byte kgpu_token, position / start as running sums per
// sentence.  stream: the 24-byte records go out with non-temporal stores (three 8-byte movnti per token; `out` 8-byte aligned) -- the expansion of a large
// call writes hundreds of megabytes nobody reads back soon, and an ordinary store first READS every line it is about to overwrite: a third of the
// expansion's memory traffic, and the expansion is a memory-bandwidth job.  The caller fences (expand_fence) before it publishes the result.
inline void expand_tokens(const kgpu_token8 *in, const uint64_t *tok_offsets, const uint32_t *first, uint64_t n, kgpu_token *out, bool stream) {
    const uint64_t base = tok_offsets[0];
#if defined(__x86_64__)
    if (stream && ((uintptr_t)out & 7u) == 0) {
        for (uint64_t s = 0; s < n; ++s) {
            uint32_t pos = first[2 * s], st = first[2 * s + 1];
            for (uint64_t k = tok_offsets[s] - base, e = tok_offsets[s + 1] - base; k < e; ++k) {
                const uint32_t p = in[k].packed, chars = KGPU_T8_CHARS(p), bytes = KGPU_T8_BYTES(p);
                long long *o = (long long *)(out + k);
                _mm_stream_si64(o, (long long)((uint64_t)(uint32_t)in[k].id | ((uint64_t)KGPU_T8_CLS(p) << 32)));
                _mm_stream_si64(o + 1, (long long)((uint64_t)pos | ((uint64_t)st << 32)));
                _mm_stream_si64(o + 2, (long long)((uint64_t)(st + chars) | ((uint64_t)bytes << 32)));
                pos += bytes; st += chars;
            }
        }
        return;
    }
#endif
    for (uint64_t s = 0; s < n; ++s) {
        uint32_t pos = first[2 * s], st = first[2 * s + 1];
        for (uint64_t k = tok_offsets[s] - base, e = tok_offsets[s + 1] - base; k < e; ++k) {
            const uint32_t p = in[k].packed, chars = KGPU_T8_CHARS(p), bytes = KGPU_T8_BYTES(p);
            out[k] = kgpu_token{in[k].id, KGPU_T8_CLS(p), pos, st, st + chars, bytes};
            pos += bytes; st += chars;
        }
    }
}
inline void expand_fence() {
#if defined(__x86_64__)
    _mm_sfence();
#endif
}
constexpr uint64_t EXPAND_STREAM_MIN_TOKENS = 32768;   // from this many tokens (768 KB of records) in one chunk on: non-temporal stores
inline bool expand_stream_wanted(uint64_t tokens) {       // (KGPU_EXPAND_STREAM=0 / 1: measurement, forces ordinary / non-temporal stores)
    static const int mode = [] { const char *e = getenv("KGPU_EXPAND_STREAM"); return e ? atoi(e) : -1; }();
    return mode >= 0 ? mode != 0 : tokens >= EXPAND_STREAM_MIN_TOKENS;
}

struct WorkerPool {
    std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; std::vector<std::thread> th;
    std::mutex done_mu; std::condition_variable done_cv;
    unsigned start();                         // threads running (0: none could be created)
    void submit(std::function<void()> f);
    void wait_zero(std::atomic<int> &counter); // until the tasks counted there are through
    void task_done(std::atomic<int> &counter); // a task's last statement
};
WorkerPool &workers();
struct TestHooks { bool no_small_calls = false, legacy_host_path = false, plain_leaves = false, byte_trie = false; uint64_t chunk_bytes = 4ull << 20, chunk_sents = 16384, depth = 12, multi_chunk_sents = 0; };
TestHooks test_hooks();  // test-only environment hooks, read once per process (or per call under KGPU_TEST_HOOKS_REREAD)
void parallel_copy(void *dst, const void *src, size_t bytes);
bool is_pinned_host(const void *p);
// the pooled contexts of a dictionary (one per call in flight)
int pool_get(kgpu_dict *d, kgpu_ctx **c);
void pool_put(kgpu_dict *d, kgpu_ctx *c);
int tokenize_device_impl(kgpu_ctx *c, const uint8_t *d_utf8, const uint64_t *d_offsets, uint64_t n, uint64_t total_bytes,
                         kgpu_token *d_tokens, kgpu_token8 *d_tokens8, uint32_t *d_first, uint8_t *status8, uint64_t *toff8, uint64_t token_capacity,
                         uint64_t *d_tok_offsets, uint8_t *d_status, const char *who);

}  // namespace kgpu

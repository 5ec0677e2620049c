// tools/ubench/lockfree_combiner.h -- a LEAD, not part of the library: the small-call combiner (kgpu_api.cpp: small_call_combined) with NO lock on the way in.
// Written at the end of round 5 (profiles/experiments/r05_leads_for_round6.md, 1); exercised on the CPU by combiner_sim.cpp next to the locked forms, its
// rates on the GPU box unmeasured.
//
// The shipped combiner takes one lock per caller to join the batch being assembled; a batch's two dozen followers are released by one wake-up and reach that
// lock together (r05_callers_cpu.txt: 5.7-17.5 us of CPU per call at 128 threads even as a spinlock).  Here a batch is one 64-bit word
//
//     [63] closed | [62:31] generation | [30:23] callers in | [22:15] sentences in | [14:0] bytes in
//
// and joining is ONE compare-and-swap on it (same generation, not closed, room for my sentences and bytes) followed by a store of my request pointer into the
// slot the old count names.  The leader closes with a fetch-or -- the count it returns is final --, waits until those slots are filled (a joiner is between
// its compare-and-swap and its store for nanoseconds), launches, writes every caller's result, stores the generation into the batch's `done` word and wakes
// the sleepers on it.  Batches live in a fixed ring and are handed back by the last caller out (`left` reaches the final count): no allocation, no shared_ptr.
// A caller that finds no batch to join (none open, closed, full, another generation) leads a new one; with the ring exhausted it launches alone.
//
// Launch is a template parameter: `void operator()(Req *const *reqs, size_t n)` serves the n requests (the library: small_call()).
#pragma once
#include <atomic>
#include <climits>
#include <cstddef>
#include <cstdint>
#include <ctime>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace kgpu_lead {

struct CombinerLimits { uint32_t max_callers = 128, max_sentences = 128, max_bytes = 16384; unsigned window_us = 12; };

template <class Req, class Launch>
class LockFreeCombiner {
    static constexpr int RING = 64, SLOTS = 128;
    static constexpr uint64_t CLOSED = 1ull << 63;
    static uint64_t pack(uint32_t gen, uint32_t count, uint32_t n, uint32_t bytes) { return ((uint64_t)gen << 31) | ((uint64_t)count << 23) | ((uint64_t)n << 15) | bytes; }
    static uint32_t gen_of(uint64_t s) { return (uint32_t)((s & ~CLOSED) >> 31); }
    static uint32_t count_of(uint64_t s) { return (uint32_t)(s >> 23) & 0xFF; }
    static uint32_t n_of(uint64_t s) { return (uint32_t)(s >> 15) & 0xFF; }
    static uint32_t bytes_of(uint64_t s) { return (uint32_t)s & 0x7FFF; }
    struct alignas(64) Batch {
        std::atomic<uint64_t> state{0};
        alignas(64) std::atomic<uint32_t> done{0};   // futex word: the generation whose results are written
        std::atomic<uint32_t> left{0}, busy{0};
        alignas(64) std::atomic<Req *> slots[SLOTS];
    };
    Batch ring_[RING];
    alignas(64) std::atomic<uint64_t> open_{0};      // (ring index + 1) << 32 | generation; 0: nothing to join
    alignas(64) std::atomic<uint32_t> next_gen_{0};
    alignas(64) std::atomic<int> callers_{0};
    Launch launch_;
    CombinerLimits lim_;

    static void relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    static void futex_wait(std::atomic<uint32_t> *w, uint32_t expected) { syscall(SYS_futex, (uint32_t *)w, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0); }
    static void futex_wake_all(std::atomic<uint32_t> *w) { syscall(SYS_futex, (uint32_t *)w, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }
    static long long now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1000000ll + t.tv_nsec / 1000; }

    void leave(Batch *b) {   // (after `done`: the state is closed, its count final)
        const uint32_t cnt = count_of(b->state.load(std::memory_order_acquire));
        if (b->left.fetch_add(1, std::memory_order_acq_rel) + 1 == cnt) {   // the last one out hands the batch back
            for (uint32_t i = 0; i < cnt; ++i) b->slots[i].store(nullptr, std::memory_order_relaxed);
            b->left.store(0, std::memory_order_relaxed);
            b->busy.store(0, std::memory_order_release);
        }
    }
    Batch *claim(int *idx) {
        for (int i = 0; i < RING; ++i) {
            uint32_t z = 0;
            if (ring_[i].busy.load(std::memory_order_relaxed) == 0 && ring_[i].busy.compare_exchange_strong(z, 1, std::memory_order_acquire)) { *idx = i; return &ring_[i]; }
        }
        return nullptr;
    }

public:
    std::atomic<uint64_t> batches{0}, joined{0}, alone{0};   // statistics
    explicit LockFreeCombiner(Launch l = Launch(), CombinerLimits lim = CombinerLimits()) : launch_(l), lim_(lim) {
        for (auto &b : ring_) for (auto &s : b.slots) s.store(nullptr, std::memory_order_relaxed);
    }
    int callers() const { return callers_.load(std::memory_order_relaxed); }

    // `n` sentences, `bytes` bytes in this request; returns when the request has been served
    void call(Req &me, uint32_t n, uint32_t bytes) {
        struct In { std::atomic<int> &c; In(std::atomic<int> &c_) : c(c_) { c.fetch_add(1, std::memory_order_acq_rel); } ~In() { c.fetch_sub(1, std::memory_order_acq_rel); } } in(callers_);
        if (const uint64_t o = open_.load(std::memory_order_acquire)) {
            Batch *b = &ring_[(o >> 32) - 1];
            const uint32_t gen = (uint32_t)o;
            uint64_t st = b->state.load(std::memory_order_relaxed);
            while (gen_of(st) == gen && !(st & CLOSED) && count_of(st) < lim_.max_callers && count_of(st) < (uint32_t)SLOTS &&
                   n_of(st) + n <= lim_.max_sentences && bytes_of(st) + bytes <= lim_.max_bytes) {
                if (b->state.compare_exchange_weak(st, st + pack(0, 1, n, bytes), std::memory_order_acq_rel, std::memory_order_relaxed)) {
                    b->slots[count_of(st)].store(&me, std::memory_order_release);
                    for (uint32_t d; (d = b->done.load(std::memory_order_acquire)) != gen;) futex_wait(&b->done, d);   // my result is written before the word is
                    joined.fetch_add(1, std::memory_order_relaxed);
                    leave(b);
                    return;
                }
            }
        }
        // lead
        int idx = 0;
        Batch *b = claim(&idx);
        if (!b) { Req *one = &me; launch_(&one, 1); alone.fetch_add(1, std::memory_order_relaxed); return; }   // (more batches in flight than the ring holds)
        uint32_t gen = next_gen_.fetch_add(1, std::memory_order_relaxed) + 1;
        if (gen == 0) gen = next_gen_.fetch_add(1, std::memory_order_relaxed) + 1;   // 0 is "no generation"
        b->slots[0].store(&me, std::memory_order_relaxed);
        b->state.store(pack(gen, 1, n, bytes), std::memory_order_release);
        const uint64_t mine = ((uint64_t)(idx + 1) << 32) | gen;
        open_.store(mine, std::memory_order_release);   // (a batch another leader still holds open stays its leader's: it receives no more joiners and closes itself)
        if (lim_.window_us && callers_.load(std::memory_order_acquire) > 1) {
            const long long t0 = now_us();
            for (;;) {
                for (int k = 0; k < 16; ++k) relax();
                const uint64_t st = b->state.load(std::memory_order_acquire);
                if (count_of(st) >= lim_.max_callers || n_of(st) >= lim_.max_sentences || bytes_of(st) + 256 > lim_.max_bytes) break;   // full
                if ((int)count_of(st) >= callers_.load(std::memory_order_acquire)) break;                                               // everyone who is here is in
                if (now_us() - t0 >= (long long)lim_.window_us) break;
            }
        }
        const uint64_t fin = b->state.fetch_or(CLOSED, std::memory_order_acq_rel);
        const uint32_t cnt = count_of(fin);
        { uint64_t e = mine; open_.compare_exchange_strong(e, 0, std::memory_order_acq_rel); }
        Req *reqs[SLOTS];
        reqs[0] = &me;
        for (uint32_t i = 1; i < cnt; ++i) {
            Req *r;
            for (unsigned spins = 0; !(r = b->slots[i].load(std::memory_order_acquire)); ++spins) { relax(); if ((spins & 1023) == 1023) sched_yield(); }
            reqs[i] = r;
        }
        launch_(reqs, cnt);
        batches.fetch_add(1, std::memory_order_relaxed);
        b->done.store(gen, std::memory_order_release);
        if (cnt > 1) futex_wake_all(&b->done);
        leave(b);
    }
};

}  // namespace kgpu_lead

// tools/ubench/combiner_sim.cpp -- the small-call combiner WITHOUT a device: T threads make one-sentence calls in a closed loop, a "launch" is a sleep of L us
// after which the leader writes every caller's result (a function of its input, checked by the caller).  What is compared is the way INTO a batch:
//   mutex       the combiner as it was until round 5 (leader / follower under a pthread mutex, one futex wake-all per batch)
//   spin        the shipped form: the same under a test-and-test-and-set spinlock with backoff (a copy of kgpu_api.cpp's SpinLock)
//   spinpark    bounded spinning, then asleep on the lock word (spin_then_park_lock.h)
//   lockfree    one compare-and-swap on the batch's state word (lockfree_combiner.h)
// per form: calls/s, p50 / p99 latency, CPU microseconds per call (getrusage of the process: on a box with more hardware threads than the cgroup's CPU quota
// this is the number that decides whether the quota stops the process), callers per launch.  Every result is verified; exit code 1 on a wrong one.
// build: g++ -std=c++17 -O2 -pthread tools/ubench/combiner_sim.cpp -o combiner_sim ; run: ./combiner_sim [threads=128] [calls per thread=2000] [launch us=80]
// (-fsanitize=thread works: the requests' fields are plain, the combiners' ordering is what keeps the report clean)
#include "lockfree_combiner.h"
#include "spin_then_park_lock.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <sys/resource.h>
#include <thread>
#include <vector>

struct Req { uint64_t in = 0, out = 0; };
static uint64_t expected_of(uint64_t in) { return in * 0x9E3779B97F4A7C15ull + 12345; }

struct SleepLaunch {
    unsigned us;
    void operator()(Req *const *reqs, size_t n) const {
        timespec ts{0, (long)us * 1000};
        if (us) nanosleep(&ts, nullptr);
        for (size_t i = 0; i < n; ++i) reqs[i]->out = expected_of(reqs[i]->in);
    }
};

struct ShippedSpinLock {   // kgpu_api.cpp: struct SpinLock, as of round 5
    std::atomic<uint32_t> v{0};
    void lock() {
        for (unsigned spins = 0, backoff = 1;;) {
            if (v.load(std::memory_order_relaxed) == 0) {
                if (v.exchange(1, std::memory_order_acquire) == 0) return;
                for (unsigned k = 0; k < backoff; ++k) __builtin_ia32_pause();
                if (backoff < 32) backoff *= 2;
            }
            __builtin_ia32_pause();
            if (++spins >= 2048) { sched_yield(); spins = 0; }
        }
    }
    void unlock() { v.store(0, std::memory_order_release); }
};

// The shipped combiner's structure (kgpu_api.cpp: small_call_combined) around a lock of choice
template <class Lock>
struct LockedCombiner {
    struct Batch { std::vector<Req *> reqs; bool closed = false; std::atomic<uint32_t> done{0}; };
    alignas(64) Lock mu;
    std::shared_ptr<Batch> open;
    alignas(64) std::atomic<int> callers{0};
    SleepLaunch launch;
    unsigned window_us = 12;
    std::atomic<uint64_t> batches{0}, joined{0}, alone{0};
    explicit LockedCombiner(SleepLaunch l) : launch(l) {}
    static long long now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1000000ll + t.tv_nsec / 1000; }
    void call(Req &me, uint32_t, uint32_t) {
        struct In { std::atomic<int> &c; In(std::atomic<int> &c_) : c(c_) { c.fetch_add(1); } ~In() { c.fetch_sub(1); } } in(callers);
        std::shared_ptr<Batch> mine;
        {
            std::unique_lock<Lock> l(mu);
            std::shared_ptr<Batch> b = open;
            if (b && !b->closed && b->reqs.size() < 128) {
                b->reqs.push_back(&me);
                l.unlock();
                while (b->done.load(std::memory_order_acquire) == 0) syscall(SYS_futex, (uint32_t *)&b->done, FUTEX_WAIT_PRIVATE, 0, nullptr, nullptr, 0);
                joined.fetch_add(1, std::memory_order_relaxed);
                return;
            }
            mine = std::make_shared<Batch>();
            mine->reqs.reserve(128);
            mine->reqs.push_back(&me);
            open = mine;
        }
        if (window_us && callers.load() > 1) {
            const long long t0 = now_us();
            for (;;) {
                for (int k = 0; k < 16; ++k) __builtin_ia32_pause();
                if (now_us() - t0 >= (long long)window_us) break;
                std::lock_guard<Lock> g(mu);
                if (mine->reqs.size() >= 128 || (int)mine->reqs.size() >= callers.load()) break;
            }
        }
        {
            std::lock_guard<Lock> g(mu);
            mine->closed = true;
            if (open == mine) open.reset();
        }
        launch(mine->reqs.data(), mine->reqs.size());
        batches.fetch_add(1, std::memory_order_relaxed);
        const bool had = mine->reqs.size() > 1;
        mine->done.store(1, std::memory_order_release);
        if (had) syscall(SYS_futex, (uint32_t *)&mine->done, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    }
};

static double cpu_seconds() {
    rusage r; getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6;
}

template <class Combiner>
static bool run(const char *name, Combiner &cb, int threads, int calls) {
    std::vector<std::vector<float>> lat((size_t)threads);
    std::atomic<uint32_t> go{0};
    std::atomic<int> ready{0};
    std::atomic<uint64_t> wrong{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            lat[(size_t)t].reserve((size_t)calls);
            ready.fetch_add(1);
            while (go.load(std::memory_order_acquire) == 0) syscall(SYS_futex, (uint32_t *)&go, FUTEX_WAIT_PRIVATE, 0, nullptr, nullptr, 0);
            for (int k = 0; k < calls; ++k) {
                Req r;
                r.in = (uint64_t)t * 1000003u + (uint64_t)k;
                const auto t0 = std::chrono::steady_clock::now();
                cb.call(r, 1, 120);
                lat[(size_t)t].push_back(std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count());
                if (r.out != expected_of(r.in)) wrong.fetch_add(1);
            }
        });
    while (ready.load() < threads) std::this_thread::sleep_for(std::chrono::microseconds(100));
    const double c0 = cpu_seconds();
    const auto w0 = std::chrono::steady_clock::now();
    go.store(1, std::memory_order_release);
    syscall(SYS_futex, (uint32_t *)&go, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    for (auto &x : th) x.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(), cpu = cpu_seconds() - c0;
    std::vector<float> all;
    for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    const double n = (double)all.size();
    const uint64_t launches = cb.batches.load() + cb.alone.load();
    printf("%-9s %4d threads x %d: %8.0f k calls/s  p50 %6.0f  p99 %7.0f us  CPU %6.2f us per call (%4.1f CPUs busy)  %5.1f callers per launch  wrong %llu\n", name, threads, calls,
           n / wall / 1e3, all[all.size() / 2], all[(size_t)(n * 0.99)], cpu * 1e6 / n, cpu / wall, n / (double)(launches ? launches : 1), (unsigned long long)wrong.load());
    return wrong.load() == 0;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 128, calls = argc > 2 ? atoi(argv[2]) : 2000;
    const unsigned launch_us = argc > 3 ? (unsigned)atoi(argv[3]) : 80;
    const char *only = argc > 4 ? argv[4] : "";
    auto want = [&](const char *n) { return !*only || std::string(only) == n; };
    printf("the combiner without a device: a launch = %u us of sleep; %u hardware threads here\n", launch_us, std::thread::hardware_concurrency());
    bool ok = true;
    if (want("mutex")) { LockedCombiner<std::mutex> c{SleepLaunch{launch_us}}; ok = run("mutex", c, threads, calls) && ok; }
    if (want("spin")) { LockedCombiner<ShippedSpinLock> c{SleepLaunch{launch_us}}; ok = run("spin", c, threads, calls) && ok; }
    if (want("spinpark")) { LockedCombiner<kgpu::SpinThenParkLock> c{SleepLaunch{launch_us}}; ok = run("spinpark", c, threads, calls) && ok; }
    if (want("lockfree")) {
        auto c = std::make_unique<kgpu_lead::LockFreeCombiner<Req, SleepLaunch>>(SleepLaunch{launch_us});
        ok = run("lockfree", *c, threads, calls) && ok;
        printf("          (lockfree: %llu joined, %llu launches led, %llu alone with the ring exhausted)\n", (unsigned long long)c->joined.load(), (unsigned long long)c->batches.load(), (unsigned long long)c->alone.load());
    }
    return ok ? 0 : 1;
}

// Stress of the small-call combiner's lock on the CPU.  tests/test_host_lock_cpu.py cuts `struct SpinLock { ... };` out of kanpyo_amd/csrc/kgpu_api.cpp as it
// stands and passes it in with -include: T threads x N increments of a PLAIN counter (and of a two-word invariant) under the lock, also with a holder that
// sleeps inside the critical section now and then.  Prints "ok ..." or "FAIL ...".  Built with -fsanitize=thread where the toolchain has it: the counter is
// not atomic, so a lock that let two threads in is a reported race as well as a wrong sum.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

static bool run(const char *name, int threads, int iters, int nap_every) {
    alignas(64) static SpinLock lock;
    static uint64_t counter, a, b;
    counter = a = b = 0;
    bool torn = false;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            for (int i = 0; i < iters; ++i) {
                std::lock_guard<SpinLock> g(lock);
                if (a != b) torn = true;
                ++a;
                if (nap_every && (i + t) % nap_every == 0) std::this_thread::sleep_for(std::chrono::microseconds(60));   // the waiters run through their yields
                ++counter;
                ++b;
            }
        });
    for (auto &x : th) x.join();
    const uint64_t want = (uint64_t)threads * (uint64_t)iters;
    const bool ok = counter == want && a == want && b == want && !torn && lock.v.load() == 0;
    printf("%s %s: %d threads x %d, counter %llu of %llu, lock word %u\n", ok ? "ok" : "FAIL", name, threads, iters, (unsigned long long)counter,
           (unsigned long long)want, (unsigned)lock.v.load());
    return ok;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 20000;
    bool ok = run("every thread at once", threads, iters, 0);
    ok = run("a holder that naps", threads, iters / 40 ? iters / 40 : 1, 97) && ok;
    return ok ? 0 : 1;
}

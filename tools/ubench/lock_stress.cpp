// Stress of spin_then_park_lock.h on the CPU (g++ -std=c++17 -O2 -pthread [-fsanitize=thread] tools/ubench/lock_stress.cpp -o lock_stress; ./lock_stress 32 20000): T threads x N increments of a PLAIN counter (and of a two-word invariant) under the lock, with the shipped
// spin bound, with a bound of 0 (every contended acquisition parks on the futex: the slow path and its wake-ups) and with a holder that sleeps inside the
// critical section now and then (waiters run out of spins while the lock is held).  Prints "ok <sum>" or "FAIL ...".  Built with -fsanitize=thread where the
// toolchain has it: the counter is not atomic, so a lock that let two threads in is a reported race as well as a wrong sum.
#include "spin_then_park_lock.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

template <class Lock>
static bool run(const char *name, int threads, int iters, int nap_every) {
    alignas(64) static Lock lock;
    static uint64_t counter, a, b;
    counter = a = b = 0;
    bool torn = false;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            for (int i = 0; i < iters; ++i) {
                std::lock_guard<Lock> g(lock);
                if (a != b) torn = true;
                ++a;
                if (nap_every && (i + t) % nap_every == 0) std::this_thread::sleep_for(std::chrono::microseconds(60));   // longer than the spin bound
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
    bool ok = run<kgpu::SpinThenParkLock>("shipped spin bound", threads, iters, 0);
    ok = run<kgpu::SpinThenParkLockT<0>>("no spinning (always the futex path)", threads, iters / 4, 0) && ok;
    ok = run<kgpu::SpinThenParkLockT<64>>("short spin bound", threads, iters, 0) && ok;
    ok = run<kgpu::SpinThenParkLock>("a holder that naps", threads, iters / 40, 97) && ok;
    return ok ? 0 : 1;
}

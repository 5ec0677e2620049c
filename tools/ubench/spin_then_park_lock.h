// tools/ubench/spin_then_park_lock.h -- a LEAD, not part of the library: the small-call combiner's lock (kgpu_api.cpp: SpinLock) with BOUNDED spinning.  Written at the
// end of round 5, after the spinlock was measured: it passed this directory's lock_stress.cpp (also under -fsanitize=thread) and, built into the library,
// tests/test_gpu_concurrent.py on the GPU box -- but the round's GPU minutes ended before its rates were measured, so the library keeps the measured lock.
// To try it: include this header in kgpu_api.cpp, make Combiner::mu a kgpu::SpinThenParkLock, run tools/probes/r05_probe27.sh and bench.py's callers legs.
//
// Held for a push_back and two additions (tens of nanoseconds), taken by every caller -- and by a whole batch's followers at the same instant, when the leader's
// one wake-up releases them into their next calls.  A pthread mutex puts each of them to sleep and wakes it again through the kernel: measured with 128 callers,
// 40-48 us of (system) CPU per call in the lock alone -- more CPU than a 16-CPU cgroup quota grants, so the group spent most of each 100 ms period throttled
// (profiles/experiments/r05_callers_cpu.txt).  So the waiters spin first (test-and-test-and-set, pause, a lost exchange backs off for up to 32 pauses): two dozen
// of them pass the lock on in a few microseconds.  The spinning is bounded (SPIN_PAUSES, 20-30 us): a holder that lost its CPU -- to the scheduler, or to the
// quota, which stops the group's CPUs one by one -- is waited for asleep on the lock word (0 free, 1 held, 2 held with sleepers: U. Drepper, "Futexes are
// tricky", the third mutex), not by burning the period's remaining budget.  lock() / unlock(): std::lock_guard and std::unique_lock work on it.
#pragma once
#include <atomic>
#include <climits>
#include <cstdint>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace kgpu {

template <unsigned SPIN_PAUSES>
struct SpinThenParkLockT {
    std::atomic<uint32_t> v{0};
    void lock() {
        for (unsigned pauses = 0, backoff = 1; pauses < SPIN_PAUSES; ++pauses) {
            if (v.load(std::memory_order_relaxed) == 0) {
                uint32_t free_word = 0;
                if (v.compare_exchange_strong(free_word, 1, std::memory_order_acquire, std::memory_order_relaxed)) return;
                for (unsigned k = 0; k < backoff; ++k) cpu_relax();
                pauses += backoff;
                if (backoff < 32) backoff *= 2;
            }
            cpu_relax();
        }
        // taken with 2: my unlock wakes the next sleeper, if there is one (and costs one needless wake-up call if there is none)
        while (v.exchange(2, std::memory_order_acquire) != 0) syscall(SYS_futex, (uint32_t *)&v, FUTEX_WAIT_PRIVATE, 2, nullptr, nullptr, 0);
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
using SpinThenParkLock = SpinThenParkLockT<1500>;

}  // namespace kgpu

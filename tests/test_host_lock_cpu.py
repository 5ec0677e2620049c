"""The small-call combiner's lock (kgpu_api.cpp: struct SpinLock -- test-and-test-and-set with backoff, a few yields, then a futex sleep on the word) stressed on the CPU, no
device: the struct is cut out of the source AS IT STANDS and compiled into tests/c_abi/lock_stress.cpp.  The combiner serves the reference's call shape --
tokenize(&self) from many threads, one sentence per call (src/tokenizer.rs:16, src/bin/kanpyo.rs:106-126); its parity under load is
tests/test_gpu_concurrent.py's business, this test is about the lock letting exactly one thread in."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "lock_stress.cpp")
API = os.path.join(ROOT, "kanpyo_amd", "csrc", "kgpu_api.cpp")


def shipped_lock(d):
    s = open(API).read()
    i = s.index("struct SpinLock {")
    j = s.index("\n};", i) + 3
    assert "alignas(64) SpinLock mu;" in s, "the combiner no longer uses SpinLock: point this test at its lock"
    h = os.path.join(d, "shipped_lock.h")
    with open(h, "w") as f:
        f.write("#include <atomic>\n#include <cstdint>\n#include <sched.h>\n#include <linux/futex.h>\n#include <sys/syscall.h>\n#include <unistd.h>\n" + s[i:j] + "\n")
    return h


def build(d, flags, name):
    exe = os.path.join(d, name)
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-pthread", *flags, "-include", shipped_lock(d), SRC, "-o", exe], capture_output=True, text=True)
    return r, exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("threads,iters", [(32, 20000), (128, 3000)])
def test_lock_counts_every_increment(threads, iters):
    with tempfile.TemporaryDirectory() as d:
        r, exe = build(d, ["-Wall", "-Werror"], "lock_stress")
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe, str(threads), str(iters)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "FAIL" not in r.stdout and r.stdout.count("ok ") == 2, r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_lock_under_thread_sanitizer():
    """The counter under the lock is a plain variable: ThreadSanitizer reports any pair of threads the lock let in together (skipped where the
    toolchain cannot build or start a -fsanitize=thread binary)."""
    with tempfile.TemporaryDirectory() as d:
        r, exe = build(d, ["-fsanitize=thread"], "lock_stress_tsan")
        if r.returncode != 0:
            pytest.skip("g++ cannot build with -fsanitize=thread here")
        r = subprocess.run([exe, "8", "4000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
        if "FATAL: ThreadSanitizer" in r.stderr:
            pytest.skip("the ThreadSanitizer runtime does not start here: " + r.stderr[:200])
        assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[:2000]

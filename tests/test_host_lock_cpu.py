"""The small-call combiner's lock (kanpyo_amd/csrc/kgpu_lock.h: bounded spinning, then a futex sleep on the lock word) stressed on the CPU, no device:
tests/c_abi/lock_stress.cpp includes the very header kgpu_api.cpp is built with.  The combiner serves the reference's call shape -- tokenize(&self) from
many threads, one sentence per call (src/tokenizer.rs:16, src/bin/kanpyo.rs:106-126); its parity under load is tests/test_gpu_concurrent.py's business."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "lock_stress.cpp")


def build(flags, out):
    return subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-pthread", *flags, SRC, "-o", out], capture_output=True, text=True)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_lock_counts_every_increment():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "lock_stress")
        r = build(["-Wall", "-Werror"], exe)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe, "32", "20000"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout + r.stderr
        assert r.stdout.count("ok ") == 4, r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_lock_under_thread_sanitizer():
    """The counter under the lock is a plain variable: ThreadSanitizer reports any pair of threads the lock let in together (skipped where the
    toolchain cannot build or run a -fsanitize=thread binary)."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "lock_stress_tsan")
        r = build(["-fsanitize=thread"], exe)
        if r.returncode != 0:
            pytest.skip("g++ cannot build with -fsanitize=thread here")
        r = subprocess.run([exe, "8", "4000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
        if r.returncode not in (0, 1, 66) or "FATAL: ThreadSanitizer" in r.stderr:
            pytest.skip("the ThreadSanitizer runtime does not start here: " + r.stderr[:200])
        assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[:2000]

"""tools/ubench/combiner_sim.cpp: the small-call combiner without a device (a launch = a sleep), the shipped way into a batch next to the two leads kept
under tools/ubench/ (bounded spinning then a futex sleep; joining by one compare-and-swap, lockfree_combiner.h).  Not product code: this test only keeps the
leads compiling and correct -- every caller must get the result computed for ITS request, in every form (the reference's call shape: tokenize(&self) from many
threads, src/tokenizer.rs:16)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "combiner_sim.cpp")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_every_form_serves_every_caller():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "combiner_sim")
        r = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Werror", SRC, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for args in (("48", "400", "20"), ("300", "60", "40", "lockfree")):   # (the second: more batches in flight than the ring holds)
            r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and "wrong 0" in r.stdout, r.stdout + r.stderr
            assert all(line.rstrip().endswith("wrong 0") for line in r.stdout.splitlines() if " threads x " in line), r.stdout

"""The oracle under AddressSanitizer + UBSan (the reference is safe Rust, its restatement is C):
the golden vectors, the edge cases and a slice of the differential fuzz run against oracle/libkanpyo_oracle_asan.so in a
child process with the sanitizer runtime preloaded; any report fails the test.  CPU only."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
from conftest import fixture_dict_parts, load_golden
from kanpyo_amd.dict import Dict
from kanpyo_amd import synth
from kanpyo_amd.tokenizer import pack_sentences
from oracle import oracle
assert oracle.lib()._name.endswith("libkanpyo_oracle_asan.so")
d = Dict.from_parts(**fixture_dict_parts())
o = oracle.OracleTokenizer.from_dict(d)
for case in load_golden("fixture_tokens.json")["cases"]:
    got, _ = o.tokenize(case["input"])
    assert [[int(x) for x in list(t)[:5]] for t in got] == [t[:5] for t in case["tokens"]], case["input"]
sd = synth.build_dict(6000, seed=5)
o2 = oracle.OracleTokenizer.from_dict(sd.dict)
sents = synth.make_corpus(sd, 300, 3, "cfg2") + synth.make_corpus(sd, 40, 4, "cfg3") + ["", "\x00", "ア" * 1500, "1" * 1025, "\U00020000"]
utf8, offs = pack_sentences(sents)
r1 = o2.tokenize_batch(utf8, offs, 1)
r3 = o2.tokenize_batch(utf8, offs, 3)
assert np.array_equal(r1.tokens, r3.tokens) and np.array_equal(r1.offsets, r3.offsets)
blob = oracle.index_build(sorted({{w for w in sd.surfaces[:500]}}, key=lambda s: s.encode()))
assert oracle.da_search(blob, "never-a-key") is None
print("sanitized ok", len(sents), int(r1.offsets[-1]))
"""


def test_oracle_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    from oracle import oracle

    oracle.build(asan=True)
    golden = os.path.join(ROOT, "tests", "golden", "fixture_tokens.json")
    assert os.path.exists(golden)
    env = dict(os.environ, KORC_ASAN="1", LD_PRELOAD=libasan + (":" + libubsan if os.path.exists(libubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]

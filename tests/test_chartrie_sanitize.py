"""The character-level trie builder (kanpyo_amd/csrc/kgpu_chartrie.cpp: host code that re-indexes an UNTRUSTED double array at
kgpu_dict_create) under AddressSanitizer + UBSan: random dictionaries over 1- to 4-byte alphabets, queries against the byte-level search,
a 65 535-character dictionary (the builder must decline), truncated and garbage arrays (it must not read out of bounds).  The file is
compiled on its own with g++ (no HIP), loaded in a child process with the sanitizer runtime preloaded.  CPU only."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import ctypes as C, random, struct, sys
sys.path.insert(0, {root!r})
import numpy as np
from oracle import oracle
L = C.CDLL({lib!r})
f = L.kgpu_debug_chartrie_search
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]

def char_search(blob, queries):
    text = b"".join(queries) or b"\0"
    offs = np.zeros(len(queries) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(q) for q in queries])
    cap = 16 * len(text) + 16
    out = np.zeros((cap, 2), dtype=np.uint32); ooff = np.zeros(len(queries) + 1, dtype=np.uint64); info = np.zeros(3, dtype=np.uint64)
    tb = np.frombuffer(text, dtype=np.uint8)
    rc = f(blob, len(blob), tb.ctypes.data, offs.ctypes.data, len(queries), out.ctypes.data, cap, ooff.ctypes.data, info.ctypes.data)
    return rc, [[(int(a), int(b)) for a, b in out[int(ooff[i]):int(ooff[i + 1])]] for i in range(len(queries))], info

def byte_search(blob, q):
    (n,) = struct.unpack_from("<Q", blob, 0)
    da = np.frombuffer(blob, dtype="<i4", count=2 * n, offset=8).reshape(n, 2)
    p, out = 1, []
    for i, ch in enumerate(q):
        prev = p; p = int(da[prev, 0]) + ch
        if not (0 <= p < n) or da[p, 1] != prev: break
        ah = int(da[p, 0])
        if 0 <= ah < n and da[ah, 1] == p and da[ah, 0] < 0: out.append((-int(da[ah, 0]), i + 1))
    return out

rng = random.Random(3)
alphabets = ["あいうえおかき", "abcAB01 .", "äöüßλμЖ", "\U00020BB7\U0001F600\U00029E3D", "東京都￿"]
checked = 0
for trial in range(25):
    alpha = "".join(rng.sample(alphabets, rng.randint(1, len(alphabets))))
    keys = set()
    for _ in range(rng.choice([1, 8, 80, 800])):
        keys.add("".join(rng.choice(alpha) for _ in range(rng.randint(1, rng.choice([2, 5, 9])))))
    keys = sorted(keys, key=lambda s: s.encode())
    blob = oracle.index_build(keys)
    qs = [(rng.choice(keys) + "".join(rng.choice(alpha + "xyz𝒵") for _ in range(rng.randint(0, 3)))).encode() for _ in range(120)] + [b""]
    rc, got, info = char_search(blob, qs)
    assert rc == 0 and info[2] == 1
    for q, g in zip(qs, got):
        assert g == byte_search(blob, q), (trial, q)
        checked += 1
    # truncated and scrambled arrays: whatever comes back, no out-of-bounds access (the sanitizer is the assertion)
    (n,) = struct.unpack_from("<Q", blob, 0)
    cut = struct.pack("<Q", max(2, n // 2)) + blob[8:8 + 8 * max(2, n // 2)] + struct.pack("<Q", 0)
    char_search(cut, qs[:20])
    junk = bytearray(blob)
    for _ in range(40):
        k = 8 + 4 * rng.randrange(2 * n)
        junk[k:k + 4] = struct.pack("<i", rng.choice([0, 1, -1, 2, n - 1, n, n + 7, -n, 0x7FFFFFFF, -0x80000000, rng.randrange(-n, 2 * n)]))
    da = np.frombuffer(bytes(junk), dtype="<i4", count=2 * n, offset=8).reshape(n, 2)
    p = int(da[1, 1])  # kgpu_dict_create refuses an array whose root is some node's child (child edges could loop): same pre-check here
    if not (1 <= p < n and p != 1 and 0 <= 1 - int(da[p, 0]) <= 255 and da[p, 0] >= 0):
        char_search(bytes(junk), qs[:20])
chars = [chr(c) for c in range(0x20, 0xD800)] + [chr(c) for c in range(0xE000, 0xFFFE)] + [chr(c) for c in range(0x10000, 0x10000 + 3000)]
rc, _, info = char_search(oracle.index_build(sorted(chars, key=lambda s: s.encode())), [b"a"])
assert rc == 0 and info[2] == 0
print("sanitized ok", checked)
"""


def test_chartrie_builder_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    lib = str(tmp_path / "libkgpu_chartrie_asan.so")
    src = os.path.join(ROOT, "kanpyo_amd", "csrc", "kgpu_chartrie.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-I", os.path.join(ROOT, "include"), src, "-o", lib], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    from oracle import oracle

    oracle.build()
    env = dict(os.environ, LD_PRELOAD=libasan + (":" + libubsan if os.path.exists(libubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lib=lib)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]

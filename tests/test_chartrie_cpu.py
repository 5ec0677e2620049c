"""The character-level copy of the trie (kanpyo_amd/csrc/kgpu_chartrie.cpp) against the byte-level common-prefix search
(reference kanpyo-dict/src/trie/da.rs:155-182, restated in oracle/pyref.py): same (id, byte length) matches for every query.
Host only: the library's test hook builds the array from an index.dict blob and runs the kernels' walk on the CPU."""
import ctypes as C
import random
import struct

import numpy as np

from kanpyo_amd import _lib
from kanpyo_amd.dict import index_table_build


def byte_level_search(blob: bytes, q: bytes):
    (n,) = struct.unpack_from("<Q", blob, 0)
    da = np.frombuffer(blob, dtype="<i4", count=2 * n, offset=8).reshape(n, 2)
    base, check = da[:, 0], da[:, 1]
    p, out = 1, []
    for i, ch in enumerate(q):
        prev = p
        p = int(base[prev]) + ch
        if not (0 <= p < n) or check[p] != prev:
            break
        ah = int(base[p])
        if 0 <= ah < n and check[ah] == p and base[ah] < 0:
            out.append((-int(base[ah]), i + 1))
    return out


def char_level_search(blob: bytes, queries):
    L = _lib.lib()
    f = L.kgpu_debug_chartrie_search
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    text = b"".join(queries) or b"\0"
    offs = np.zeros(len(queries) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    cap = 16 * len(text) + 16
    out = np.zeros((cap, 2), dtype=np.uint32)
    ooff = np.zeros(len(queries) + 1, dtype=np.uint64)
    info = np.zeros(3, dtype=np.uint64)
    tb = np.frombuffer(text, dtype=np.uint8)
    rc = f(blob, len(blob), tb.ctypes.data, offs.ctypes.data, len(queries), out.ctypes.data, cap, ooff.ctypes.data, info.ctypes.data)
    assert rc == 0
    assert info[2] == 1, "the character-level array was not built"
    return [[(int(a), int(b)) for a, b in out[int(ooff[i]) : int(ooff[i + 1])]] for i in range(len(queries))], info


ALPHABETS = [
    "あいうえおかきくけこさしすせそたちつてと",          # 3-byte characters
    "abcdeABCDE01239 .",                                  # 1-byte
    "äöüßéèçñøåλμπΩЖдя",                                   # 2-byte
    "𠮷𩸽😀😁🙂𝒳𝒴",                                         # 4-byte (beyond the BMP table)
    "東京都大阪府名古屋市中区港区￿",                      # incl. U+FFFF (the kernels' "not in the table" value)
]


def random_keys(rng, n, alphabet, maxlen):
    keys = set()
    for _ in range(20 * n):  # (a small alphabet has fewer short strings than asked for)
        if len(keys) >= n:
            break
        k = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, maxlen)))
        keys.add(k)
        if rng.random() < 0.5 and len(k) < maxlen:  # prefixes of each other: several matches per query
            keys.add(k + rng.choice(alphabet))
    return sorted(keys, key=lambda s: s.encode())


def test_same_matches_as_the_byte_level_walk():
    rng = random.Random(20260928)
    for trial in range(40):
        alphabet = "".join(rng.sample(ALPHABETS, rng.randint(1, len(ALPHABETS))))
        keys = random_keys(rng, rng.choice([1, 5, 60, 600]), alphabet, rng.choice([2, 4, 9]))
        blob = index_table_build(keys)
        other = "ゃゅょxyzЩ𝒵"  # characters in no key
        queries = []
        for _ in range(300):
            if rng.random() < 0.5:
                q = rng.choice(keys) + "".join(rng.choice(alphabet + other) for _ in range(rng.randint(0, 4)))
            else:
                q = "".join(rng.choice(alphabet + other) for _ in range(rng.randint(0, 8)))
            queries.append(q.encode())
        queries += [b"", keys[0].encode(), keys[-1].encode()]
        got, info = char_level_search(blob, queries)
        for q, g in zip(queries, got):
            assert g == byte_level_search(blob, q), (trial, q.decode(), g)
        assert info[1] <= len(set("".join(keys)))


def test_duplicate_keys_and_a_large_dictionary():
    """Duplicates are adjacent in the sorted keyword list (index.rs:16-38): one trie id, the walk reports it once."""
    rng = random.Random(7)
    base = random_keys(rng, 20000, ALPHABETS[0] + ALPHABETS[4], 6)
    keys = sorted(base + rng.sample(base, 500), key=lambda s: s.encode())
    blob = index_table_build(keys)
    queries = [(rng.choice(base) + rng.choice(base)).encode() for _ in range(2000)]
    got, info = char_level_search(blob, queries)
    assert info[0] < 8 * len(base)  # the array stays compact: a few slots per key
    for q, g in zip(queries, got):
        assert g == byte_level_search(blob, q)


def test_what_the_copy_cannot_represent_falls_back_to_the_byte_walk():
    """65 535 or more distinct characters in the keys (16-bit codes, 0xFFFF = none), or a corrupt array whose child edges loop: the builder
    says no (the dictionary is then walked byte by byte on the device, as in rounds 1-2) instead of looping or running out of memory."""
    chars = [chr(c) for c in range(0x20, 0xD800)] + [chr(c) for c in range(0xE000, 0xFFFE)] + [chr(c) for c in range(0x10000, 0x10000 + 3000)]
    assert len(chars) >= 0xFFFF
    blob = index_table_build(sorted(chars, key=lambda s: s.encode()))
    L = _lib.lib()
    f = L.kgpu_debug_chartrie_search
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]

    def built(b):
        offs, out, ooff, info = np.zeros(2, dtype=np.uint64), np.zeros((4, 2), dtype=np.uint32), np.zeros(2, dtype=np.uint64), np.zeros(3, dtype=np.uint64)
        offs[1] = 1
        t = np.frombuffer(b"a", dtype=np.uint8)
        assert f(b, len(b), t.ctypes.data, offs.ctypes.data, 1, out.ctypes.data, 4, ooff.ctypes.data, info.ctypes.data) == 0
        return int(info[2])

    assert built(blob) == 0
    # a loop: the root (slot 1, base 2) has child 'a' at slot 99, whose base 0 makes slot 1 ITS child by byte 0x01 -- kgpu_dict_create checks the
    # root's parent before it hands the array to the builder (every slot has one parent, so a loop can only close through the root)
    usable = L.kgpu_debug_char_trie_usable
    usable.restype = C.c_int
    usable.argtypes = [C.c_void_p, C.c_size_t]
    n = 200
    da = np.zeros((n, 2), dtype="<i4")
    da[1] = (2, 99)
    da[99] = (0, 1)
    looped = struct.pack("<Q", n) + da.tobytes() + struct.pack("<Q", 0)
    assert usable(looped, len(looped)) == 0
    da[1] = (2, 0)  # ... the same array without the loop is fine
    ok = struct.pack("<Q", n) + da.tobytes() + struct.pack("<Q", 0)
    assert usable(ok, len(ok)) == 1 and built(ok) == 1
    assert usable(blob, len(blob)) == 0

"""Differential fuzz of the two CPU restatements (oracle/kanpyo_oracle.c vs the naive oracle/pyref.py) -- the only
token-level cross-check available while the reference cannot be built: >= 10 000 random sentences
over random small dictionaries with negative costs, missing unk entries, every invoke/group permutation,
duplicate surfaces, nested prefixes, non-BMP characters and dead ends.  CPU only."""
import numpy as np
import pytest

from kanpyo_amd.dict import Dict
from oracle import oracle, pyref

ALPHABETS = [
    "あいうえお", "アイウエオカ", "日本語辞書形態素", "abcXYZ", "0123", "、。 ", "\U00020000\U0002A6D6",  # non-BMP -> category of table[0]
]


def _random_dict(rng, n_cat):
    alpha = "".join(ALPHABETS)
    n_words = int(rng.integers(1, 60))
    words = set()
    while len(words) < n_words:
        src = rng.choice(len(ALPHABETS))
        w = "".join(rng.choice(list(ALPHABETS[src]), size=int(rng.integers(1, 5))))
        words.add(w)
        if rng.random() < 0.4 and len(w) < 6:  # nested prefix
            words.add(w + str(rng.choice(list(alpha))))
    records = []
    for w in sorted(words, key=lambda s: s.encode("utf-8")):
        for _ in range(int(rng.choice([1, 1, 1, 2, 3, 6]))):  # duplicate surfaces
            records.append(w)
    n_ctx = int(rng.integers(1, 7))
    morphs = np.stack([rng.integers(0, n_ctx, len(records)), rng.integers(0, n_ctx, len(records)),
                       rng.integers(-3000, 9000, len(records))], axis=1)
    matrix = rng.integers(-4000, 4000, size=n_ctx * n_ctx)
    cat = np.zeros(65536, dtype=np.uint8)
    for k, a in enumerate(ALPHABETS[:-1]):
        for ch in a:
            cat[ord(ch)] = k % n_cat
    invoke = rng.integers(0, 2, n_cat).astype(np.uint8)
    group = rng.integers(0, 2, n_cat).astype(np.uint8)
    unk_map, unk_morphs, nxt = {}, [], 1
    for c in range(n_cat):
        if rng.random() < 0.75:  # some categories have no unk entry: dead ends, unreachable EOS
            cnt = int(rng.integers(1, 4))
            unk_map[c] = (nxt, cnt)
            for _ in range(cnt):
                unk_morphs.append((int(rng.integers(0, n_ctx)), int(rng.integers(0, n_ctx)), int(rng.integers(-2000, 12000))))
            nxt += cnt
    return Dict.from_parts(records, morphs, n_ctx, n_ctx, matrix, [f"C{c}" for c in range(n_cat)], cat, invoke, group, unk_map, unk_morphs)


def _sentence(rng):
    parts = []
    for _ in range(int(rng.integers(0, 7))):
        a = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))]
        parts.append("".join(rng.choice(list(a), size=int(rng.integers(1, 6)))))
    return "".join(parts)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_pyref_and_c_oracle_agree(seed):
    rng = np.random.default_rng(seed)
    total = 0
    for _ in range(50):
        d = _random_dict(rng, int(rng.integers(1, 6)))
        parts = (d.index_dict, d.connection_dict, d.morph_dict, d.unk_dict, d.char_category, d.invoke_list, d.group_list)
        o = oracle.OracleTokenizer(*parts)
        p = pyref.PyDict(*parts)
        for _ in range(55):
            s = _sentence(rng)
            exp = pyref.tokenize(p, s)
            got, _ctr = o.tokenize(s)
            assert [tuple(int(x) for x in t) for t in got] == exp, (seed, s)
            total += 1
    assert total >= 2500  # x 4 seeds = 11 000 sentences


@pytest.mark.parametrize("seed", [11, 12])
def test_pyref_and_c_oracle_agree_on_keys_of_every_utf8_width(seed):
    """synth.width_case (1- to 4-byte characters and U+FFFF anywhere in a key): the dictionaries the gpu tests use to pin the device's
    character-level trie walk -- here the two CPU restatements of the byte-level walk (trie/da.rs:155-182) against each other."""
    import random

    from kanpyo_amd import synth

    rng = random.Random(seed)
    total = 0
    for _ in range(12):
        d, sents = synth.width_case(rng)
        parts = (d.index_dict, d.connection_dict, d.morph_dict, d.unk_dict, d.char_category, d.invoke_list, d.group_list)
        o = oracle.OracleTokenizer(*parts)
        p = pyref.PyDict(*parts)
        for s in sents[:40]:
            s = s[:60]
            exp = pyref.tokenize(p, s)
            got, _ctr = o.tokenize(s)
            assert [tuple(int(x) for x in t) for t in got] == exp, (seed, s)
            total += 1
    assert total >= 200

"""Randomised GPU-vs-oracle parity, where the driver runs it (`pytest -m gpu`): tools/fuzz_parity.py's generators with fixed
seeds and a fixed number of rounds.  The dense little dictionaries reach what the IPADIC-shaped corpora never do -- every
shape of the Viterbi step (P <= 8 / <= 16 / <= 32 / beyond, T beyond one pass, T > 127; reference src/lattice.rs:121-140) and
more than eight prefixes at one position (src/lattice.rs:24-38) -- under every launch chain: pool shapes, plain-leaf layout,
pool kernel off (long-sentence / HBM-scratch kernels only).  Integer work: exact equality."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POOLS = ["0", "8:2:64", "16:4:32", "40:4:32", "40:4:40", "40:4:48", "40:8:20", "80:8:20", "80:10:64", "160:16:64", "80:8:20,160:4:64", "24:3:48"]
WINDOWS = ["0", "8", "10", "12", "16", "32"]  # KiB of LDS of the windowed kernel behind the pools ("0": off, the general kernel takes its place)


@pytest.fixture(scope="module")
def libs():
    from kanpyo_amd import _lib

    assert _lib.lib().kgpu_device_count() > 0, "no HIP device: the gpu tests need an MI355X"
    from oracle import oracle

    oracle.build()
    return _lib, oracle


def _same(tok, orc, sentences, what):
    from kanpyo_amd.tokenizer import pack_sentences

    utf8, offs = pack_sentences(sentences)
    exp = orc.tokenize_batch(utf8, offs, 16)
    got_t, got_off, status = tok.tokenize_packed(utf8, offs)
    assert not status.any(), what
    assert np.array_equal(got_off, exp.offsets), f"{what}: per-sentence token counts differ"
    if not np.array_equal(got_t, exp.tokens):
        bad = int(np.nonzero(got_t != exp.tokens)[0][0])
        s = int(np.searchsorted(exp.offsets, bad, side="right") - 1)
        raise AssertionError(f"{what}: token {bad} (sentence {s}: {sentences[s]!r}) gpu {got_t[bad]} oracle {exp.tokens[bad]}")
    return len(sentences)


@pytest.mark.parametrize("seed", [9001, 4242, 777, 31337])
def test_dense_dictionaries_every_sweep_shape(libs, seed, monkeypatch):
    """Twelve dense dictionaries per seed, each under a launch chain drawn from the same seed."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    rng = random.Random(seed)
    total = 0
    for k in range(12):
        pool, window_kib = rng.choice(POOLS), rng.choice(WINDOWS)
        plain = rng.random() < 0.25
        monkeypatch.setenv("KGPU_POOL", pool)
        monkeypatch.setenv("KGPU_WINDOW", window_kib)
        if plain:
            monkeypatch.setenv("KGPU_PLAIN_LEAVES", "1")
        else:
            monkeypatch.delenv("KGPU_PLAIN_LEAVES", raising=False)
        d, sents = synth.dense_case(rng)
        tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
        total += _same(tok, orc, sents, f"seed {seed} round {k} pool={pool} window={window_kib} plain={plain}")
    assert total > 0


@pytest.mark.parametrize("seed", [20260929, 1234])
def test_keys_of_every_utf8_width_random_chains(libs, seed, monkeypatch):
    """Sixteen dictionaries per seed over alphabets of 1- to 4-byte characters (synth.width_case): the character-level copy of the trie
    (kgpu_chartrie.cpp) against the reference's byte-level walk (trie/da.rs:155-182) under random launch chains, the windowed kernel on or
    off, and -- one in five -- with the byte-level walk itself on the device (KGPU_BYTE_TRIE)."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    rng = random.Random(seed)
    total = 0
    for k in range(16):
        pool, window_kib = rng.choice(POOLS), rng.choice(WINDOWS)
        byte_trie = rng.random() < 0.2
        monkeypatch.setenv("KGPU_POOL", pool)
        monkeypatch.setenv("KGPU_WINDOW", window_kib)
        monkeypatch.setenv("KGPU_BYTE_TRIE", "1" if byte_trie else "0")
        d, sents = synth.width_case(rng)
        tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
        total += _same(tok, orc, sents, f"seed {seed} round {k} pool={pool} window={window_kib} byte_trie={byte_trie}")
    assert total > 0


@pytest.mark.parametrize("seed,nkeys", [(9001, 20000), (4242, 6000), (777, 60000)])
def test_mixed_corpora_random_chains(libs, seed, nkeys, monkeypatch):
    """IPADIC-shaped dictionaries of random size, shuffled mixes of cfg 2 / cfg 3 / cfg 5 text and edge sentences, three
    batches per chain (the reservation estimate and the optional-launch heuristics adapt between calls)."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    rng = random.Random(seed)
    sd = synth.build_dict(nkeys, seed=rng.randrange(1 << 30))
    orc = oracle.OracleTokenizer.from_dict(sd.dict)
    rng2 = random.Random(seed + 1)   # (its own stream: the chains and corpora above stay what they were before these two knobs existed)
    for k in range(4):
        pool, window_kib = rng.choice(POOLS), rng.choice(WINDOWS)
        team, first = rng2.choice(["-1", "0", "2", "2"]), rng2.choice(["1024", "0", "64", "300"])   # round 5: the team form / chains that start with the windowed kernel
        if k == 3:   # the last chain of every seed: the shipped plan (no KGPU_POOL: the runtime then picks the pool's shape by the chain, round 5)
            monkeypatch.delenv("KGPU_POOL", raising=False)
            pool = "auto"
        else:
            monkeypatch.setenv("KGPU_POOL", pool)
        monkeypatch.setenv("KGPU_WINDOW", window_kib)
        monkeypatch.setenv("KGPU_WINDOW_TEAM", team)
        monkeypatch.setenv("KGPU_WINDOW_FIRST", first)
        tok = Tokenizer(sd.dict)
        for j in range(3):
            _same(tok, orc, synth.mixed_case(sd, rng, sizes=(1, 5, 50, 120, 700, 4096)), f"seed {seed} chain {k}.{j} pool={pool} window={window_kib} team={team} first={first}")


def test_default_chain_many_small_dictionaries(libs):
    """The shipped launch plan (no environment overrides) over twenty dense dictionaries and their small calls."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    rng = random.Random(20260928)
    for k in range(20):
        d, sents = synth.dense_case(rng)
        tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
        _same(tok, orc, sents, f"default chain round {k}")
        _same(tok, orc, sents[:7], f"default chain round {k} (small call)")

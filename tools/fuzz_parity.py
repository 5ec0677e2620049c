#!/usr/bin/env python
"""Randomised GPU-vs-oracle parity sweep (GPU box): random dictionary sizes, corpus mixes, pool / long-kernel
shapes, batch sizes.  usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first: shares one HIP runtime)

from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import pack_sentences
from oracle import oracle

from kanpyo_amd.dict import Dict


def dense_dict(rng):
    """A tiny alphabet with many duplicate records per surface: buckets of 1..200 predecessors, up to ~150 targets per
    position -- every shape of the sweep step (P <= 8, <= 16, <= 32, beyond; T beyond one pass; T > 127) and the parked-
    match overflow (more than 8 prefixes at a position)."""
    nr = np.random.default_rng(rng.randrange(1 << 30))
    alpha = "あいうえおか"[: rng.choice([2, 3, 6])]
    words = set()
    for _ in range(rng.choice([20, 80, 300])):
        words.add("".join(nr.choice(list(alpha), size=int(nr.integers(1, rng.choice([3, 5, 12]))))))
    recs = []
    for w in sorted(words, key=lambda x: x.encode()):
        recs += [w] * int(nr.choice([1, 1, 2, 3, 9, rng.choice([17, 33, 70])]))
    nctx = rng.choice([1, 3, 40])
    morphs = np.stack([nr.integers(0, nctx, len(recs)), nr.integers(0, nctx, len(recs)), nr.integers(-2000, 9000, len(recs))], axis=1)
    cat = np.zeros(65536, dtype=np.uint8)
    for ch in alpha:
        cat[ord(ch)] = 1
    unk = {0: (1, 1), 1: (2, rng.choice([1, 3]))}
    um = [(0, 0, 5000)] + [(int(nr.integers(0, nctx)), int(nr.integers(0, nctx)), int(nr.integers(1000, 9000))) for _ in range(unk[1][1])]
    d = Dict.from_parts(recs, morphs, nctx, nctx, nr.integers(-3000, 3000, nctx * nctx), ["DEFAULT", "H"], cat,
                        np.array([0, rng.choice([0, 1])], dtype=np.uint8), np.array([1, rng.choice([0, 1])], dtype=np.uint8), unk, um)
    sents = ["".join(nr.choice(list(alpha + "xy"), size=int(nr.integers(1, rng.choice([8, 40, 120])))))
             for _ in range(rng.choice([1, 7, 128, 129, 700, 3000]))]
    return d, sents


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
rounds = sentences = 0
EDGE = ["", "あ", "ア" * 700, "a" * 300, "𠮷野家で𩸽", "すもももももももものうち", "　　", "1234567890" * 40, "。" * 65]
while time.time() < t_end:
    nkeys = rng.choice([6000, 12000, 20000, 60000])
    pool = rng.choice(["0", "8:2:64", "16:4:32", "40:4:32", "40:4:32", "40:8:20", "80:8:20", "80:10:64", "160:16:64", "80:8:20,160:4:64", "24:3:48"])
    long_kib = rng.choice(["0", "4", "12", "32", "160"])
    os.environ["KGPU_POOL"], os.environ["KGPU_LONG"] = pool, long_kib
    if rng.random() < 0.4:  # a dense little dictionary: wide buckets, many targets
        dd, mix = dense_dict(rng)
        tok, orc = Tokenizer(dd), oracle.OracleTokenizer.from_dict(dd)
        print(f"[{time.time() - (t_end - budget):6.1f}s] dense dictionary pool={pool} long={long_kib} n={len(mix)}", flush=True)
        utf8, offs = pack_sentences(mix)
        exp = orc.tokenize_batch(utf8, offs, 16)
        got_t, got_off, status = tok.tokenize_packed(utf8, offs)
        if not (np.array_equal(got_off, exp.offsets) and np.array_equal(got_t, exp.tokens) and not status.any()):
            print(f"MISMATCH dense pool={pool} long={long_kib} n={len(mix)}")
            sys.exit(1)
        rounds += 1
        sentences += len(mix)
        continue
    sd = synth.build_dict(nkeys, seed=rng.randrange(1 << 30))
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    print(f"[{time.time() - (t_end - budget):6.1f}s] keys={nkeys} pool={pool} long={long_kib}", flush=True)
    for _ in range(3):
        mix = []
        mix += synth.make_corpus(sd, rng.choice([1, 5, 50, 120, 700, 4096, 9000]), rng.randrange(1 << 30), "cfg2")
        if rng.random() < 0.7:
            mix += synth.make_corpus(sd, rng.choice([3, 100, 600]), rng.randrange(1 << 30), "cfg3")
        if rng.random() < 0.3:
            mix += synth.make_corpus(sd, rng.choice([1, 4]), rng.randrange(1 << 30), "cfg5")
        mix += rng.sample(EDGE, rng.randrange(len(EDGE)))
        rng.shuffle(mix)
        utf8, offs = pack_sentences(mix)
        exp = orc.tokenize_batch(utf8, offs, 16)
        print(f"    n={len(mix)} bytes={int(offs[-1])} ...", end="", flush=True)
        got_t, got_off, status = tok.tokenize_packed(utf8, offs)
        print(" done", flush=True)
        ok = np.array_equal(got_off, exp.offsets) and np.array_equal(got_t, exp.tokens) and not status.any()
        if not ok:
            print(f"MISMATCH keys={nkeys} pool={pool} long={long_kib} n={len(mix)}")
            sys.exit(1)
        rounds += 1
        sentences += len(mix)
print(f"fuzz ok: {rounds} batches, {sentences} sentences, bit-exact")

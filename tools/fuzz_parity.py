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

os.environ["KGPU_TEST_HOOKS_REREAD"] = "1"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
rounds = sentences = 0
while time.time() < t_end:
    nkeys = rng.choice([6000, 12000, 20000, 60000])
    pool = rng.choice(["0", "8:2:64", "16:4:32", "40:4:32", "40:4:32", "40:8:20", "80:8:20", "80:10:64", "160:16:64", "80:8:20,160:4:64", "24:3:48", "auto", "auto", "auto"])
    window_kib = rng.choice(["0", "8", "10", "12", "12", "16", "24"])  # the windowed kernel behind the pools (0 = off: the general kernel takes its place)
    os.environ["KGPU_WINDOW"] = window_kib
    if pool == "auto":   # the shipped plan: no KGPU_POOL at all -- only then does the runtime pick the pool's shape by the chain (two wavefronts on 20 KB in front of a windowed launch)
        os.environ.pop("KGPU_POOL", None)
    else:
        os.environ["KGPU_POOL"] = pool
    os.environ["KGPU_WINDOW_TEAM"] = rng.choice(["-1", "-1", "0", "2", "2"])     # the two-wavefronts-per-sentence form of window-first chains: by the load / never / always
    os.environ["KGPU_WINDOW_FIRST"] = rng.choice(["1024", "1024", "0", "64", "300"])  # average bytes per sentence from which a chain starts with the windowed kernel
    os.environ["KGPU_BYTE_TRIE"] = "1" if rng.random() < 0.15 else "0"  # (read at dictionary creation: KGPU_TEST_HOOKS_REREAD below)
    if rng.random() < 0.4:  # a dense little dictionary: wide buckets, many targets -- or keys of every UTF-8 width
        dd, mix = synth.dense_case(rng) if rng.random() < 0.6 else synth.width_case(rng)
        tok, orc = Tokenizer(dd), oracle.OracleTokenizer.from_dict(dd)
        print(f"[{time.time() - (t_end - budget):6.1f}s] dense / width dictionary byte_trie={os.environ['KGPU_BYTE_TRIE']} pool={pool} window={window_kib} team={os.environ['KGPU_WINDOW_TEAM']} first={os.environ['KGPU_WINDOW_FIRST']} n={len(mix)}", flush=True)
        utf8, offs = pack_sentences(mix)
        exp = orc.tokenize_batch(utf8, offs, 16)
        got_t, got_off, status = tok.tokenize_packed(utf8, offs)
        if not (np.array_equal(got_off, exp.offsets) and np.array_equal(got_t, exp.tokens) and not status.any()):
            print(f"MISMATCH dense pool={pool} window={window_kib} n={len(mix)}")
            sys.exit(1)
        rounds += 1
        sentences += len(mix)
        continue
    sd = synth.build_dict(nkeys, seed=rng.randrange(1 << 30))
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    print(f"[{time.time() - (t_end - budget):6.1f}s] keys={nkeys} byte_trie={os.environ['KGPU_BYTE_TRIE']} pool={pool} window={window_kib} team={os.environ['KGPU_WINDOW_TEAM']} first={os.environ['KGPU_WINDOW_FIRST']}", flush=True)
    for _ in range(3):
        mix = synth.mixed_case(sd, rng)
        utf8, offs = pack_sentences(mix)
        exp = orc.tokenize_batch(utf8, offs, 16)
        print(f"    n={len(mix)} bytes={int(offs[-1])} ...", end="", flush=True)
        got_t, got_off, status = tok.tokenize_packed(utf8, offs)
        print(" done", flush=True)
        ok = np.array_equal(got_off, exp.offsets) and np.array_equal(got_t, exp.tokens) and not status.any()
        if not ok:
            print(f"MISMATCH keys={nkeys} pool={pool} window={window_kib} n={len(mix)}")
            sys.exit(1)
        rounds += 1
        sentences += len(mix)
print(f"fuzz ok: {rounds} batches, {sentences} sentences, bit-exact")

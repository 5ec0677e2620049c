#!/usr/bin/env python
"""Call latency of the host-buffer entry point (kgpu_tokenize_batch) for n = 1, 64, 1024, 4096 sentences per call.
usage (GPU box): [KGPU_HOST_CHUNK_SENTS=..] python tools/latency_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences
sd = synth.build_dict(); corpus = synth.make_corpus(sd, 4096, 1, "cfg2"); tok = Tokenizer(sd.dict)
utf8, offs = pack_sentences(corpus)
cap = int(offs[-1]) + 4096
out = (np.empty(cap, dtype=TOKEN_DTYPE), np.empty(4097, dtype=np.uint64), np.empty(4096, dtype=np.uint8))
for n in (1, 8, 64, 1024, 4096):
    o = offs[: n + 1].copy(); u = utf8[: int(o[-1])]
    for _ in range(30): tok.tokenize_packed(u, o, out=out)
    ts = []
    for _ in range(300 if n < 1024 else 60):
        t0 = time.perf_counter(); tok.tokenize_packed(u, o, out=out); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"n={n:5d}: median {ts[len(ts)//2]*1e6:8.1f} us  p10 {ts[len(ts)//10]*1e6:8.1f}  p90 {ts[len(ts)*9//10]*1e6:8.1f}   {n/ts[len(ts)//2]/1e6:7.3f} M sentences/s")

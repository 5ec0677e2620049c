#!/usr/bin/env python
"""One large host call (400k cfg 2 sentences), pageable buffers: python tools/e2e_quick.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences
sd = synth.build_dict(); tok = Tokenizer(sd.dict)
u1, o1 = pack_sentences(synth.make_corpus(sd, 100000, 1, "cfg2"))
reps = 4
u = np.tile(u1, reps); o = np.concatenate([[0]] + [o1[1:] + k * int(o1[-1]) for k in range(reps)]).astype(np.uint64)
n = len(o) - 1
out = (np.empty(int(o[-1]) // 2 + n, dtype=TOKEN_DTYPE), np.empty(n + 1, dtype=np.uint64), np.empty(n, dtype=np.uint8))
out[0].view(np.uint8)[::4096] = 0
tok.tokenize_packed(u, o, out=out)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); tok.tokenize_packed(u, o, out=out); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"large call: {n / ts[len(ts)//2] / 1e6:.1f} M sentences/s (median of 7; best {n / ts[0] / 1e6:.1f})")
u0, o0 = u1[: int(o1[4096])], o1[:4097]
for _ in range(20): tok.tokenize_packed(u0, o0, out=out)
ts = []
for _ in range(100):
    t0 = time.perf_counter(); tok.tokenize_packed(u0, o0, out=out); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"4096 per call: {ts[50]*1e6:.0f} us median, {4096 / ts[50] / 1e6:.1f} M sentences/s")

#!/usr/bin/env python
"""n = 1 call latency with the chip idle and with a light background load that keeps its clocks up: python tools/clock_probe.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences
sd = synth.build_dict(); tok = Tokenizer(sd.dict)
sents = synth.make_corpus(sd, 256, 1, "cfg2")
out = (np.empty(4096, dtype=TOKEN_DTYPE), np.empty(300, dtype=np.uint64), np.empty(300, dtype=np.uint8))
def lat(n, reps=300):
    u, o = pack_sentences(sents[:n])
    for _ in range(30): tok.tokenize_packed(u, o, out=out)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); tok.tokenize_packed(u, o, out=out); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6
print(f"idle chip: n=1 {lat(1):.1f} us, n=64 {lat(64):.1f} us", flush=True)
stop = False
def load():
    s = torch.cuda.Stream()
    x = torch.ones(1 << 22, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(50): x.mul_(1.0001)
            s.synchronize()
th = threading.Thread(target=load); th.start(); time.sleep(0.5)
print(f"background elementwise load: n=1 {lat(1):.1f} us, n=64 {lat(64):.1f} us", flush=True)
stop = True; th.join()

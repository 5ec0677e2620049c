#!/usr/bin/env python
"""Probe of the tokenize kernels on the GPU box: per-phase shader-clock breakdown (PROFILE_WORK run, one batch at a time)
and full-occupancy throughput per ablation level (several large batches in flight).
usage: [KGPU_SPLIT=..] [KGPU_POOL=..] python tools/split_probe.py [cfg2] [n_small=4096] [n_big=32768]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import PROFILE_EVENTS, PROFILE_OFF, PROFILE_WORK, DeviceContext
from kanpyo_amd.tokenizer import pack_sentences

kind = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n_small = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_big = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
sd = synth.build_dict()
sents = synth.make_corpus(sd, max(n_small, n_big), 1, kind)
tok = Tokenizer(sd.dict)
dev = torch.device("cuda", 0)


def upload(n):
    utf8, offs = pack_sentences(sents[:n])
    cap = int(offs[-1]) + n
    return (torch.from_numpy(utf8.copy()).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), n, int(offs[-1]), cap)


def outs(cap, n):
    return (torch.empty((cap, 6), dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))


def run(ctx, b, o):
    ctx.tokenize(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3], o[0].data_ptr(), b[4], o[1].data_ptr(), o[2].data_ptr())


print("config: KGPU_SPLIT=%s KGPU_POOL=%s" % (os.environ.get("KGPU_SPLIT"), os.environ.get("KGPU_POOL")))
# ---- (a) phases, one small batch at a time
b = upload(n_small)
o = outs(b[4], n_small)
ctx = DeviceContext(tok)
for mode in (PROFILE_EVENTS, PROFILE_EVENTS | PROFILE_WORK):
    ctx.set_profiling(mode)
    for _ in range(5):
        run(ctx, b, o)
        ctx.sync()
    p = ctx.profile()
    print(f"n={n_small} mode {mode}: tokenize chain {1e3 * p['tokenize_ms'] / p['launches']:.1f} us, aux {1e3 * p['aux_ms'] / p['launches']:.1f} us, deferred {p['deferred']}, redone {p['redone']}")
ph = ctx.phase_cycles()
ns = max(ph["sentences"], 1)
print("  phase ticks are s_memtime (100 MHz): x10 ns; per sentence (sweep phases of the split path: per wavefront of 4 sentences, shown /4)")
for k, v in ph.items():
    if k != "sentences":
        print(f"  {k:18s} {v / ns * 10:10.0f} ns/sentence")
ctx.set_profiling(PROFILE_OFF)
ctx.close()
# ---- (b) throughput per ablation level, 3 big batches in flight
Q = 3
bb = upload(n_big)
ctxs = [DeviceContext(tok) for _ in range(Q)]
oo = [outs(bb[4], n_big) for _ in range(Q)]
for stop, name in ((1, "load"), (2, "decode"), (3, "walk"), (4, "scan"), (5, "emit"), (6, "gather"), (7, "sweep"), (0, "all")):
    for c in ctxs:
        c.set_ablation(stop)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4 * Q):
            ctxs[i % Q].sync() if i >= Q else None
            run(ctxs[i % Q], bb, oo[i % Q])
        for c in ctxs:
            c.sync()
        dt = time.perf_counter() - t0
    print(f"stop after {name:8s}: {4 * Q * n_big / dt / 1e6:8.1f} M sentences/s   ({dt / (4 * Q) * 1e6 / n_big * 1e3:7.2f} ns/sentence)")

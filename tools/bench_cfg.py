#!/usr/bin/env python
"""Throughput of the other BASELINE configs (not the bench.py headline):
cfg3 = mixed 8-512 chars unknown-heavy, cfg5 = 2048-char documents.
usage: python tools/bench_cfg.py cfg3 20000 [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import PROFILE_EVENTS, DeviceContext
from kanpyo_amd.tokenizer import pack_sentences

kind, n = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
sd = synth.build_dict(dense=(kind == "dense"))   # "dense": the dense-lattice variant of the dictionary under cfg 2-shaped text
sents = synth.make_corpus(sd, n, 2 if kind == "cfg3" else 5 if kind == "cfg5" else 100, "cfg2" if kind == "dense" else kind)
tok = Tokenizer(sd.dict)
dev = torch.device("cuda", 0)
bs = []
for lo in range(0, n, batch):
    utf8, offs = pack_sentences(sents[lo:lo + batch])
    m = len(offs) - 1
    cap = int(offs[-1]) + m
    bs.append((torch.from_numpy(utf8.copy()).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), m, int(offs[-1]), cap))
capmax = max(b[4] for b in bs)
Q = int(os.environ.get('BENCH_Q', '4'))
ctxs = [DeviceContext(tok) for _ in range(Q)]
outs = [(torch.empty((capmax, 6), dtype=torch.int32, device=dev), torch.empty(batch + 1, dtype=torch.int64, device=dev),
         torch.empty(batch, dtype=torch.uint8, device=dev)) for _ in range(Q)]
def run():
    for i, (u, o, m, tb, cap) in enumerate(bs):
        t, to, st = outs[i % Q]
        ctxs[i % Q].tokenize(u.data_ptr(), o.data_ptr(), m, tb, t.data_ptr(), capmax, to.data_ptr(), st.data_ptr())
    return sum(c.sync() for c in ctxs)
run(); run()
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    ntok = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
nbytes = sum(b[3] for b in bs); nchars = sum(len(s) for s in sents)
print(f"{kind}: {n} sentences, {nchars/n:.0f} chars avg: {n/dt:,.0f} sentences/s, {nchars/dt/1e6:.1f} Mchar/s, {nbytes/dt/2**20:.0f} MiB/s  ({dt*1e3:.1f} ms per pass)")

#!/usr/bin/env python
"""Where a document's time goes inside the long-sentence kernel (measurement build: make -C kanpyo_amd/csrc timing;
KGPU_POOL=0 KGPU_LIB=kanpyo_amd/libkanpyo_gpu_timing.so python tools/long_timing.py [cfg5|cfg3] [n] [in_flight])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import DeviceContext
from kanpyo_amd.tokenizer import pack_sentences
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd = synth.build_dict(); sents = synth.make_corpus(sd, n, 5 if kind == "cfg5" else 2, kind)
if kind == "cfg3": sents = [x for x in sents if len(x) > 190]; n = len(sents)
tok = Tokenizer(sd.dict); dev = torch.device("cuda", 0)
utf8, offs = pack_sentences(sents); cap = int(offs[-1]) + n
du, do = torch.from_numpy(utf8.copy()).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev)
ctxs = [DeviceContext(tok) for _ in range(Q)]
outs = [(torch.empty((cap, 6), dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(Q)]
def go(reps):
    for i in range(reps * Q):
        c = ctxs[i % Q]; o = outs[i % Q]
        if i >= Q: c.sync()
        c.tokenize(du.data_ptr(), do.data_ptr(), n, int(offs[-1]), o[0].data_ptr(), cap, o[1].data_ptr(), o[2].data_ptr())
    for c in ctxs: c.sync()
go(3)
for c in ctxs: c.phase_cycles(reset=True); c.work(reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 4; go(reps); dt = time.perf_counter() - t0
tot = np.zeros(10)
for c in ctxs: tot += np.array(list(c.phase_cycles().values()), dtype=float)
names = ["decode+count", "scan", "emit", "block set-up", "targets+gather", "chain", "write-back", "global steps", "backtrace+tokens", "blocks"]
N = reps * Q * n; chars = sum(map(len, sents)) / n
print(f"{kind}: {n} sentences of {chars:.0f} chars, in flight {Q}: {N / dt:,.0f} sentences/s")
print("  shader cycles per sentence: " + ", ".join(f"{nm} {v / N:,.0f}" for nm, v in zip(names[:9], tot)) + f"; total {tot[:9].sum() / N:,.0f} = {tot[:9].sum() / N / chars:,.0f} per char")
print(f"  blocks per sentence {tot[9] / N:.1f}: per block set-up {tot[3] / tot[9]:,.0f}, targets+gather {tot[4] / tot[9]:,.0f}, chain {tot[5] / tot[9]:,.0f} ({tot[5] / N / (chars + 1):,.0f} per position), write-back {tot[6] / tot[9]:,.0f}")

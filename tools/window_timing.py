#!/usr/bin/env python
"""Where the windowed long-sentence kernel's time goes (shader clocks per character, PROFILE_WORK run): python tools/window_timing.py cfg5|cfg3 n [Q]"""
import os, sys, time
os.environ.setdefault("KGPU_WINDOW", "10")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import PROFILE_WORK, PROFILE_OFF, PROFILE_NO_T, DeviceContext
from kanpyo_amd.tokenizer import pack_sentences
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sd = synth.build_dict(); sents = synth.make_corpus(sd, n, 5 if kind == "cfg5" else 2, kind)
if kind == "cfg3": sents = [x for x in sents if len(x) > 190]
n = len(sents)
tok = Tokenizer(sd.dict); dev = torch.device("cuda", 0)
u, o = pack_sentences(sents); cap = int(o[-1]) + n
du, do = torch.from_numpy(u.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
ctxs = [DeviceContext(tok) for _ in range(Q)]
outs = [(torch.empty((cap, 6), dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(Q)]
def go(reps):
    for i in range(reps * Q):
        c, t = ctxs[i % Q], outs[i % Q]
        if i >= Q: c.sync()
        c.tokenize(du.data_ptr(), do.data_ptr(), n, int(o[-1]), t[0].data_ptr(), cap, t[1].data_ptr(), t[2].data_ptr())
    for c in ctxs: c.sync()
go(2)
torch.cuda.synchronize(); t0 = time.perf_counter(); go(3); dt = (time.perf_counter() - t0) / (3 * Q)
chars = sum(map(len, sents))
print(f"{kind}: {n} sentences of {chars / n:.0f} chars, window LDS {os.environ['KGPU_WINDOW']} KB, {Q} in flight: {n / dt:,.0f} sentences/s, {chars / dt / 1e6:.0f} Mchar/s; reruns {sum(c.profile()['window_reruns'] for c in ctxs)}")
for c in ctxs: c.set_profiling(PROFILE_WORK | PROFILE_NO_T); c.phase_cycles(reset=True); c.work(reset=True)
go(1)
ph = np.zeros(10); w = {}
for c in ctxs:
    ph += np.array(list(c.phase_cycles().values()), dtype=float)
    for k, v in c.work().items(): w[k] = w.get(k, 0) + v
names = ["prepass", "stage", "seeds", "walk", "scan", "emit", "gather", "sweep", "flush", "backtrace+tokens"]
tot = ph.sum()
print("  cycles per char: " + ", ".join(f"{nm} {v / max(w['C'], 1):.0f}" for nm, v in zip(names, ph)) + f"; total {tot / max(w['C'], 1):.0f}")
print(f"  work per sentence: " + ", ".join(f"{k} {v / max(w['sentences'], 1):.0f}" for k, v in w.items()))

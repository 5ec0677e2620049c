#!/usr/bin/env python
"""Where a sentence's time goes inside k_tokenize_pool (measurement build: make -C kanpyo_amd/csrc timing;
KGPU_LIB=kanpyo_amd/libkanpyo_gpu_timing.so python tools/step_timing.py [cfg2] [n] [in_flight])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import DeviceContext
from kanpyo_amd.tokenizer import pack_sentences
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd = synth.build_dict(); sents = synth.make_corpus(sd, n, 1, kind)
if os.environ.get("STEP_LEN"):  # only sentences of lo..hi characters (how much of the time is the spread of lengths inside a workgroup?)
    lo, hi = map(int, os.environ["STEP_LEN"].split(","))
    sents = [x for x in synth.make_corpus(sd, 40 * n, 1, kind) if lo <= len(x) <= hi][:n]
    assert len(sents) == n, len(sents)
    print("sentences of", lo, "..", hi, "characters, mean", sum(map(len, sents)) / n)
if os.environ.get("STEP_SORT"):  # what ordering the batch by length would buy (upper bound: here the host sorts)
    sents.sort(key=lambda x: len(x.encode("utf-8")))
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
torch.cuda.synchronize(); t0 = time.perf_counter(); go(6); dt = time.perf_counter() - t0
tot = np.zeros(10)
for c in ctxs: tot += np.array(list(c.phase_cycles().values()), dtype=float)
tmS, steps, slow, slowsteps, sent, pool, N, desc = tot[:8]
idle, life = tot[8], tot[9]
ph = np.zeros(7)
for c in ctxs: ph += np.array(list(c.work().values()), dtype=float)
names = ["load", "decode", "walk", "scan", "emit + tiles", "stage B (tile gather + sweep)", "backtrace+tokens"]
print("  phases (shader cycles per sentence): " + ", ".join(f"{nm} {v / N:.0f}" for nm, v in zip(names, ph)) )
print(f"n={n} in flight={Q}: {6 * Q * n / dt / 1e6:.1f} M sentences/s; per sentence (shader cycles): whole {sent / N:.0f}, "
      f"wait for pages {pool / N:.0f}, "
      f"wavefront lifetime {life / N:.0f}, of which none (finished, its workgroup still alive) {idle / N:.0f} = {idle / max(life + idle, 1) * 100:.1f} % of the slot time; "
      f"")

#!/usr/bin/env python
"""A lone cfg 5 batch (1000 documents, ONE context, one batch in flight) and 2 / 4 / 8 in flight: python tools/team_time.py   (KGPU_WINDOW_TEAM=0|2|unset)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import DeviceContext
from kanpyo_amd.tokenizer import pack_sentences
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sd = synth.build_dict(); sents = synth.make_corpus(sd, n, 5, "cfg5")
tok = Tokenizer(sd.dict); dev = torch.device("cuda", 0)
u, o = pack_sentences(sents); cap = int(o[-1]) + n
du, do = torch.from_numpy(u.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
chars = sum(map(len, sents))
for Q in (1, 2, 4, 8):
    ctxs = [DeviceContext(tok) for _ in range(Q)]
    outs = [(torch.empty((cap, 6), dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(Q)]
    def go(reps):
        for i in range(reps * Q):
            c, t = ctxs[i % Q], outs[i % Q]
            if i >= Q: c.sync()
            c.tokenize(du.data_ptr(), do.data_ptr(), n, int(o[-1]), t[0].data_ptr(), cap, t[1].data_ptr(), t[2].data_ptr())
        for c in ctxs: c.sync()
    go(3)
    torch.cuda.synchronize(); t0 = time.perf_counter(); go(6); dt = (time.perf_counter() - t0) / (6 * Q)
    print(f"KGPU_WINDOW_TEAM={os.environ.get('KGPU_WINDOW_TEAM', 'auto')}: {n} documents, {Q} in flight: {chars / dt / 1e6:.0f} Mchar/s, {dt * 1e3:.3f} ms per batch", flush=True)
    del ctxs

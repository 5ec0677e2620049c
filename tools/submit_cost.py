#!/usr/bin/env python
"""Host-side submission cost of one kgpu_tokenize_device call vs. device time (batch 4096).
usage: python tools/submit_cost.py [batch] [threads]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import DeviceContext
from kanpyo_amd.tokenizer import pack_sentences

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nthr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NB = 24
sd = synth.build_dict()
sents = synth.make_corpus(sd, batch * NB, 100, "cfg2")
tok = Tokenizer(sd.dict)
dev = torch.device("cuda", 0)
bs = []
for lo in range(0, batch * NB, batch):
    utf8, offs = pack_sentences(sents[lo:lo + batch])
    cap = int(offs[-1]) + batch
    bs.append((torch.from_numpy(utf8.copy()).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), batch, int(offs[-1]), cap))
capmax = max(b[4] for b in bs)
ctxs = [DeviceContext(tok) for _ in range(NB)]
outs = [(torch.empty((capmax, 6), dtype=torch.int32, device=dev), torch.empty(batch + 1, dtype=torch.int64, device=dev),
         torch.empty(batch, dtype=torch.uint8, device=dev)) for _ in range(NB)]

def submit(idx):
    for i in idx:
        u, o, m, tb, cap = bs[i]
        t, to, st = outs[i]
        ctxs[i].tokenize(u.data_ptr(), o.data_ptr(), m, tb, t.data_ptr(), capmax, to.data_ptr(), st.data_ptr())

def one_pass():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if nthr == 1:
        submit(range(NB))
    else:
        th = [threading.Thread(target=submit, args=(range(k, NB, nthr),)) for k in range(nthr)]
        [t.start() for t in th]; [t.join() for t in th]
    t1 = time.perf_counter()
    for c in ctxs:
        c.sync()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0

for _ in range(3):
    one_pass()
r = [one_pass() for _ in range(5)]
sub = min(x[0] for x in r); tot = min(x[1] for x in r)
print(f"batch {batch} threads {nthr}: submit {sub/NB*1e6:.1f} us/call, total {tot/NB*1e6:.1f} us/batch => {batch*NB/tot/1e6:.1f} M sentences/s")

#!/usr/bin/env python
"""Per-phase shader-clock breakdown of the LDS-resident tokenize kernel (GPU box).
usage: python tools/phase_profile.py [cfg2|cfg3] [n_sentences]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import PROFILE_EVENTS, PROFILE_WORK, DeviceContext
from kanpyo_amd.tokenizer import pack_sentences

kind = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
sd = synth.build_dict()
sents = synth.make_corpus(sd, n, 1, kind)
tok = Tokenizer(sd.dict)
utf8, offs = pack_sentences(sents)
dev = torch.device("cuda", 0)
d_utf8 = torch.from_numpy(utf8.copy()).to(dev)
d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
cap = int(offs[-1]) + n
d_tok = torch.empty((cap, 6), dtype=torch.int32, device=dev)
d_toff = torch.empty(n + 1, dtype=torch.int64, device=dev)
d_st = torch.empty(n, dtype=torch.uint8, device=dev)
ctx = DeviceContext(tok)
for mode in (PROFILE_EVENTS, PROFILE_EVENTS | PROFILE_WORK):
    ctx.set_profiling(mode)
    for _ in range(5):
        ctx.tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, int(offs[-1]), d_tok.data_ptr(), cap, d_toff.data_ptr(), d_st.data_ptr())
        ctx.sync()
    p = ctx.profile()
    print("mode", mode, "kernel ms avg", p["tokenize_ms"] / p["launches"], "aux", p["aux_ms"] / p["launches"])
ph = ctx.phase_cycles()
ns = max(ph["sentences"], 1)
tot = sum(v for k, v in ph.items() if k not in ("sentences", "spare"))
print("sentences in LDS kernel:", ns // 5, "of", n, " work:", {k: v // 5 for k, v in ctx.work().items()})
for k, v in ph.items():
    if k in ("sentences", "spare"):
        continue
    print(f"  {k:18s} {v / ns:10.0f} cycles/sentence  {100.0 * v / tot:5.1f}%")
print(f"  total              {tot / ns:10.0f} cycles/sentence (s_memtime ticks at 100 MHz => x10 ns)")

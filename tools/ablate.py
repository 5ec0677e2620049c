#!/usr/bin/env python
"""Cumulative phase cost of the LDS kernel by early-stop ablation (kgpu_ctx_set_ablation),
measured with HIP events at real occupancy.  Each value = kernel time when every
sentence stops after phase k (1 load, 2 decode, 3 walk, 4 scan, 5 emit, 6 gather,
7 sweep, 0 everything).  usage: python tools/ablate.py [cfg2] [n]"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch

    from kanpyo_amd import Tokenizer, synth
    from kanpyo_amd.device import PROFILE_EVENTS, DeviceContext
    from kanpyo_amd.tokenizer import pack_sentences

    kind, n = sys.argv[2], int(sys.argv[3])
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
    ctx.set_ablation(int(os.environ.get("ABLATE_STOP", "0")))
    ctx.set_profiling(PROFILE_EVENTS)
    for rep in range(12):
        ctx.tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, int(offs[-1]), d_tok.data_ptr(), cap, d_toff.data_ptr(), d_st.data_ptr())
        ctx.sync()
        if rep == 1:
            ctx.profile()
    p = ctx.profile()
    print("RESULT", p["tokenize_ms"] / p["launches"] * 1e3)
    sys.exit(0)

kind = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = sys.argv[2] if len(sys.argv) > 2 else "4096"
names = {1: "load", 2: "decode", 3: "walk", 4: "scan", 5: "emit", 6: "gather", 7: "sweep", 0: "all (+backtrace/tokens)"}
prev = 0.0
for k in (1, 2, 3, 4, 5, 6, 7, 0):
    env = dict(os.environ, ABLATE_STOP=str(k))
    r = subprocess.run([sys.executable, __file__, "child", kind, n], env=env, capture_output=True, text=True, timeout=300)
    us = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not us:
        print("stop", k, "failed", r.stderr[-300:]); continue
    print(f"stop after {names[k]:26s} kernel {us[0]:8.1f} us   (+{us[0] - prev:7.1f} us)")
    prev = us[0]

#!/usr/bin/env python
"""One context, one batch in flight (cfg 2 corpus in batches of 4096) and the host call latencies, for the launch plan the environment selects:
   KGPU_POOL=10:1:64 python tools/one_ctx_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, _lib, synth
_lib.lib()
from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences
from bench_engine import GpuEngine, PackedWorkload, run_job
sd = synth.build_dict(); sents = synth.make_corpus(sd, 100_000, 1, "cfg2")
tok = Tokenizer(sd.dict); dev = torch.device("cuda", 0)
u, o = pack_sentences(sents)
for q in (1, 2):
    eng = GpuEngine(tok, dev, PackedWorkload(u, o, batch=4096), queue=q, streams=0, ring=1)
    run_job(eng, 3); torch.cuda.synchronize(); t0 = time.perf_counter(); run_job(eng, 10); dt = (time.perf_counter() - t0) / 10
    prof = eng.ctxs[0].profile(reset=True)
    print(f"contexts {q}: {100_000 / dt / 1e6:.1f} M sentences/s  routing deferred {prof['deferred']} redone {prof['redone']}")
    eng.close()
u0, o0 = pack_sentences(sents[:4096]); cap = int(o0[-1]) + 4096
out = (np.empty(cap, dtype=TOKEN_DTYPE), np.empty(4097, dtype=np.uint64), np.empty(4096, dtype=np.uint8))
for n in (1, 64, 4096):
    oo = o0[: n + 1].copy(); uu = u0[: int(oo[-1])]
    for _ in range(20): tok.tokenize_packed(uu, oo, out=out)
    ts = []
    for _ in range(100):
        t1 = time.perf_counter(); tok.tokenize_packed(uu, oo, out=out); ts.append(time.perf_counter() - t1)
    ts.sort(); print(f"kgpu_tokenize_batch n={n}: median {ts[50] * 1e6:.1f} us")

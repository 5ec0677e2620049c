#!/usr/bin/env python
"""One 4096-sentence host call at a time: latency by chunking (KGPU_HOST_CHUNK_SENTS) and where the calling thread's time goes (KGPU_HOST_TRACE=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences, pinned_empty
sd = synth.build_dict(); tok = Tokenizer(sd.dict)
u, o = pack_sentences(synth.make_corpus(sd, 4096, 1, "cfg2"))
for name, alloc in (("pageable", np.empty), ("pinned", pinned_empty)):
    uu = alloc(u.shape, dtype=np.uint8); uu[:] = u
    oo = alloc(o.shape, dtype=np.uint64); oo[:] = o
    out = (alloc(int(o[-1]) // 2 + 4200, dtype=TOKEN_DTYPE), alloc(4097, dtype=np.uint64), alloc(4096, dtype=np.uint8))
    for _ in range(30): tok.tokenize_packed(uu, oo, out=out)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); tok.tokenize_packed(uu, oo, out=out); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: chunk {os.environ.get('KGPU_HOST_CHUNK_SENTS', 'default')}: median {ts[100]*1e6:.0f} us, p10 {ts[20]*1e6:.0f} us -> {4096/ts[100]/1e6:.1f} M sentences/s", flush=True)

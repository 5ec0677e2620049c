#!/usr/bin/env python
"""Which sentences come back with a non-zero status from kgpu_tokenize_batch on cfg 3?  python tools/status_repro.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import pack_sentences
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
sd = synth.build_dict(); sents = synth.make_corpus(sd, n, 2, "cfg3")
tok = Tokenizer(sd.dict)
bad = 0
for bi, lo in enumerate(range(0, n, 4096)):
    u, o = pack_sentences(sents[lo:lo + 4096])
    t, toff, st = tok.tokenize_packed(u, o)
    if st.any():
        idx = np.nonzero(st)[0]
        bad += 1
        print(f"batch {bi}: {len(idx)} sentences with status {sorted(set(st[idx].tolist()))}; first {idx[:5].tolist()} chars {[len(sents[lo + i]) for i in idx[:5]]} counts {[int(toff[i+1]-toff[i]) for i in idx[:5]]}", flush=True)
        if bad >= 3: break
print(f"{os.environ.get('TAG','')}: {bad} bad batches of {bi + 1}", flush=True)

#!/usr/bin/env python
"""The windowed kernel's two-wavefronts-per-sentence form against the oracle, forced on for everything (KGPU_POOL=0 KGPU_WINDOW_TEAM=2): cfg 5 documents, cfg 3's
mix, cfg 2, the edge sentences, the dense fuzz dictionaries -- then a lone cfg 5 batch timed in both forms.  usage (GPU box): python tools/team_check.py [quick]"""
import os, sys, time, random
os.environ.setdefault("KGPU_POOL", "0")
os.environ.setdefault("KGPU_WINDOW_TEAM", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import pack_sentences
from oracle import oracle

oracle.build()
bad = 0
def check(tok, orc, sents, label):
    global bad
    u, o = pack_sentences(sents)
    t, toff, st = tok.tokenize_packed(u, o)
    e = orc.tokenize_batch(u, o, 8)
    ok = np.array_equal(toff, e.offsets) and np.array_equal(t, e.tokens) and not st.any()
    if not ok:
        bad += 1
        k = int(np.nonzero(np.diff(toff.astype(np.int64)) != np.diff(e.offsets.astype(np.int64)))[0][0]) if not np.array_equal(toff, e.offsets) else -1
        print(f"MISMATCH {label}: first sentence with another token count {k}, status any {st.any()}", flush=True)
    r = tok.routing(reset=True)
    print(f"{'ok ' if ok else 'BAD'} {label}: {len(sents)} sentences, deferred {r['deferred'][:3]}, window reruns {r['window_reruns']}, tail reruns {r['tail_reruns']}", flush=True)

sd = synth.build_dict(20000, seed=11)
tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
check(tok, orc, synth.EDGE_SENTENCES + ["テ", "テあ", "1" * 1024, "1" * 1025, "ア" * 1500], "edge sentences")
check(tok, orc, synth.make_corpus(sd, 200, 5, "cfg5"), "cfg5 x 200 (20k dict)")
check(tok, orc, synth.make_corpus(sd, 3000, 2, "cfg3"), "cfg3 x 3000")
check(tok, orc, synth.make_corpus(sd, 5000, 1, "cfg2"), "cfg2 x 5000")
rng = random.Random(7)
for k in range(3 if len(sys.argv) > 1 else 12):
    d, sents = synth.dense_case(rng) if k % 2 == 0 else synth.width_case(rng)
    check(Tokenizer(d), oracle.OracleTokenizer.from_dict(d), sents, f"fuzz dictionary {k}")
    check(tok, orc, synth.mixed_case(sd, rng, sizes=(1, 5, 50, 700)), f"mixed batch {k}")
sdf = synth.build_dict()
tokf, orcf = Tokenizer(sdf.dict), oracle.OracleTokenizer.from_dict(sdf.dict)
docs = synth.make_corpus(sdf, 1000, 5, "cfg5")
check(tokf, orcf, docs, "cfg5 x 1000 (392k dict)")
check(tokf, orcf, synth.make_corpus(sdf, 20000, 2, "cfg3"), "cfg3 x 20000 (392k dict)")
print("mismatching checks:", bad)

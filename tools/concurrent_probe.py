#!/usr/bin/env python
"""Concurrent small calls (the reference's server shape): python tools/concurrent_probe.py [threads,...] [n per call]
KGPU_COMBINE_US / KGPU_COMBINE_LAUNCHES: the combiner's window and its cap on launches in flight."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import concurrent_callers, pack_sentences
threads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4,16,64,256").split(",")]
npc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = synth.build_dict()
tok = Tokenizer(sd.dict)
u, o = pack_sentences(synth.make_corpus(sd, 20000, 1, "cfg2"))
for t in threads:
    concurrent_callers(tok, u, o, t, 20, (npc,))
    tok.routing(reset=True)
    r = concurrent_callers(tok, u, o, t, max(100, 20000 // t), (npc,))
    rt = tok.routing()
    launches = rt["small_calls"] - rt["combined_calls"] + rt["combined_launches"]
    print(f"threads {t:4d} n {npc}: {r['sentences_per_s']/1e3:8.1f} k sentences/s  p50 {r['p50_us']:7.1f} us  p99 {r['p99_us']:8.1f}  mean {r['mean_us']:7.1f}  "
          f"sentences/launch {r['sentences']/max(launches,1):6.1f}  fallbacks {rt['small_fallbacks']}", flush=True)
    if os.environ.get("KGPU_SMALL_TRACE"):
        import ctypes as C
        from kanpyo_amd import _lib
        st = (C.c_uint64 * 8)()
        _lib.lib().kgpu_debug_small_trace(st)
        L = max(st[0], 1)
        print(f"      per launch (us): prep {st[1]/L/1e3:.1f}  launch call {st[2]/L/1e3:.1f}  poll {st[3]/L/1e3:.1f}  hand-out {st[4]/L/1e3:.1f}  ({st[0]} launches, {st[5]/L:.1f} sentences each)", flush=True)

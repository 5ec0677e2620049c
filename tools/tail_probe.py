#!/usr/bin/env python
"""How much of a 4096-sentence launch is tail?  The cfg 2 corpus as it is, with every batch sorted so that a workgroup's
wavefronts get sentences of similar length, and with the whole corpus sorted (every batch nearly uniform).
usage (GPU box): [KGPU_POOL=..] python tools/tail_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kanpyo_amd import Tokenizer, synth

sd = synth.build_dict()
corpus = synth.make_corpus(sd, bench.N_SENT, 1, "cfg2")
tok = Tokenizer(sd.dict)
dev = torch.device("cuda", 0)
G = int(os.environ.get("TAIL_G", "1024"))  # workgroups of the pool launch (4 wavefronts each with KGPU_POOL=40:4:32)
W = 4096 // G

def by_workgroup(batch):
    o = sorted(range(len(batch)), key=lambda i: len(batch[i].encode()))
    out = [None] * len(batch)
    for r, i in enumerate(o):  # rank r -> workgroup r // W, wavefront r % W -> work index g + w * G
        g, w = divmod(r, W)
        idx = g + w * G
        if idx < len(batch): out[idx] = batch[i]
    rest = [batch[i] for r, i in enumerate(o) if (r // W) + (r % W) * G >= len(batch)]
    return [x if x is not None else rest.pop() for x in out]

variants = {
    "as generated": corpus,
    "each batch arranged by workgroup": sum((by_workgroup(corpus[lo:lo + 4096]) for lo in range(0, len(corpus), 4096)), []),
    "whole corpus sorted by length": sorted(corpus, key=lambda s: len(s.encode())),
}
for name, c in variants.items():
    wl = bench.Workload([c])
    eng = bench.GpuEngine(tok, dev, wl, queue=8, streams=4, ring=1)
    bench.run_job(eng, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_job(eng, 10)
    dt = time.perf_counter() - t0
    print(f"{name:36s} {10 * len(c) / dt / 1e6:7.2f} M sentences/s")
    eng.close()

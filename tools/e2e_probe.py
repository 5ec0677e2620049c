#!/usr/bin/env python
"""End-to-end (host memory in, 24-byte records out) rate of ONE large kgpu_tokenize_batch call: python tools/e2e_probe.py [reps of the corpus] [pinned]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences, pinned_empty
reps_c = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
sd = synth.build_dict(); corpus = synth.make_corpus(sd, 100_000, 1, "cfg2")
tok = Tokenizer(sd.dict)
keep = []
if os.environ.get("PROBE_ENG"):  # what bench.py has alive when it times its large calls: eight device contexts on four torch streams (+ a 4 GiB tensor)
    from kanpyo_amd.device import DeviceContext
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    keep = [streams, [DeviceContext(tok, streams[i % 4].cuda_stream) for i in range(8)]]
    if os.environ["PROBE_ENG"] == "2":
        for c in keep[1]: c.close()
        keep = [streams]
u1, o1 = pack_sentences(corpus)
n = reps_c * len(corpus)
ua = np.tile(u1, reps_c)
oa = np.concatenate([[0]] + [o1[1:] + k * int(o1[-1]) for k in range(reps_c)]).astype(np.uint64)
alloc = pinned_empty if pinned else np.empty
u = alloc(ua.shape, dtype=np.uint8); u[:] = ua
o = alloc(oa.shape, dtype=np.uint64); o[:] = oa
cap = int(oa[-1]) // 2 + n
big = (alloc(cap, dtype=TOKEN_DTYPE), alloc(n + 1, dtype=np.uint64), alloc(n, dtype=np.uint8))
big[0].view(np.uint8)[::4096] = 0
tok.tokenize_packed(u, o, out=big)
best = 0
for _ in range(4):
    t = time.perf_counter(); tok.tokenize_packed(u, o, out=big); dt = time.perf_counter() - t
    best = max(best, n / dt)
print(f"e2e {'pinned' if pinned else 'pageable'} n={n} depth={os.environ.get('KGPU_HOST_DEPTH','8')} threads={os.environ.get('KGPU_HOST_THREADS','-')} "
      f"chunk={os.environ.get('KGPU_HOST_CHUNK_SENTS','-')} streams={os.environ.get('KGPU_STREAMS','-')}: best {best/1e6:.1f} M sentences/s", flush=True)

#!/usr/bin/env python
"""What would a kernel that lives across batches buy?  The pool kernel over batches of n sentences (static assignment: a
wavefront of a full-chip grid takes n / 4096 sentences one after the other, i.e. it works in a populated pool the way a
resident kernel's wavefronts would), Q batches in flight:   python tools/resident_probe.py <n> <Q> [reps]
(KGPU_POOL / KGPU_POOL_WG choose the pool shape and the grid.)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.device import DeviceContext
from kanpyo_amd.tokenizer import pack_sentences

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kind = os.environ.get("PROBE_KIND", "cfg2")
total = int(os.environ.get("PROBE_TOTAL", "102400"))
sd = synth.build_dict()
sents = synth.make_corpus(sd, total, 1, kind)
if os.environ.get("PROBE_LEN"):  # every sentence cut to this fraction of its length (LDS per sentence follows the length)
    f_ = float(os.environ["PROBE_LEN"])
    sents = [x[: max(4, int(len(x) * f_))] for x in sents]
tok = Tokenizer(sd.dict)
dev = torch.device("cuda", 0)
batches = []
for lo in range(0, total, n):
    u, o = pack_sentences(sents[lo:lo + n])
    batches.append((torch.from_numpy(u.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), len(o) - 1, int(o[-1])))
cap = max(b[3] + b[2] for b in batches) + 8
streams = [torch.cuda.Stream(device=dev) for _ in range(min(4, Q))]
ctxs = [DeviceContext(tok, streams[i % len(streams)].cuda_stream) for i in range(Q)]
outs = [(torch.empty((cap, 6), dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev),
         torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(Q)]


def go(passes):
    k = 0
    for _ in range(passes):
        for b in batches:
            c, o = ctxs[k % Q], outs[k % Q]
            if k >= Q:
                c.sync()
            c.tokenize(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3], o[0].data_ptr(), cap, o[1].data_ptr(), o[2].data_ptr())
            k += 1
    for c in ctxs:
        c.sync()


go(3)
if not reps:
    reps = 10
for c in ctxs:
    c.profile(reset=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
go(reps)
dt = time.perf_counter() - t0
red = sum(c.profile()["redone"][0] for c in ctxs)
dfr = sum(c.profile()["deferred"][0] for c in ctxs)
print(f"{kind} len={os.environ.get('PROBE_LEN', 'all')} lib={os.path.basename(os.environ.get('KGPU_LIB', 'default'))} n={n} Q={Q} pool={os.environ.get('KGPU_POOL', 'default')} wg={os.environ.get('KGPU_POOL_WG', '-')}: "
      f"{reps * total / dt / 1e6:.1f} M sentences/s  redone {red / (reps * total) * 100:.2f} %  deferred {dfr / (reps * total) * 100:.2f} %", flush=True)

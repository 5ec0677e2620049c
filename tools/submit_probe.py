#!/usr/bin/env python
"""Where the HOST thread's time goes while bench.py's headline job runs (cfg 2, batches of 4096, Q contexts): per batch the wait for the context's
previous batch (kgpu_ctx_sync), the enqueue itself (kgpu_tokenize_device: the launches) and the Python around them.  If the waits are short the job is
bound by the submitting thread, not by the GPU.
usage: python tools/submit_probe.py [steps] [Q] [instances]
instances > 0: instead, the plain job's rate for that many Tokenizer instances made one after the other in this process (each has its own dictionary copy in HBM and its own
shared streams): does the headline's mode (a run's rate goes with how many launches overlap) belong to the process or to the streams?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench_engine import N_SENT, GpuEngine, Workload, run_job
from kanpyo_amd import Tokenizer, _lib, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 8
instances = int(sys.argv[3]) if len(sys.argv) > 3 else 0
_lib.lib()
sd = synth.build_dict()
wl = Workload([synth.make_corpus(sd, N_SENT, seed=1, kind="cfg2")], 0, 1)
dev = torch.device("cuda", 0)
if instances > 0:
    for k in range(instances):
        tok = Tokenizer(sd.dict, device=0)
        eng = GpuEngine(tok, dev, wl, queue=Q, streams=0, ring=1)
        t_end = time.perf_counter() + 0.7
        while time.perf_counter() < t_end:
            run_job(eng, 5)
        rates = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run_job(eng, steps)
            torch.cuda.synchronize(); rates.append(steps * N_SENT / (time.perf_counter() - t0) / 1e6)
        print(f"instance {k}: " + " / ".join(f"{r:.1f}" for r in rates) + " M sentences/s")
        del eng, tok
    sys.exit(0)
tok = Tokenizer(sd.dict, device=0)
eng = GpuEngine(tok, dev, wl, queue=Q, streams=0, ring=1)
t_end = time.perf_counter() + 1.5
while time.perf_counter() < t_end:
    run_job(eng, 5)

for rep in range(3):
    # the plain job (what bench.py times)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_job(eng, steps)
    torch.cuda.synchronize(); dt_plain = time.perf_counter() - t0
    # the same job with clocks around the two library calls
    acc = {"sync": 0.0, "enqueue": 0.0}
    waits = 0
    pc = time.perf_counter
    torch.cuda.synchronize(); t0 = pc()
    nb_total = 0
    for s in range(steps):
        for b in range(eng.nb(s)):
            i = eng.seq % eng.Q
            eng.seq += 1
            if eng.occupant[i] is not None:
                a = pc()
                eng.ntok[eng.occupant[i]] = eng.ctxs[i].sync()
                d = pc() - a
                acc["sync"] += d
                waits += d > 5e-6
            d_utf8, d_off, n, total = eng.inputs[s % len(eng.inputs)][b]
            t, o, st = eng.out[s % eng.ring][b]
            a = pc()
            eng.ctxs[i].tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, total, t.data_ptr(), eng.cap, o.data_ptr(), st.data_ptr())
            acc["enqueue"] += pc() - a
            eng.occupant[i] = (s, b)
            eng.where[(s, b)] = i
            nb_total += 1
    eng.drain()
    torch.cuda.synchronize(); dt = pc() - t0
    per = 1e6 / nb_total
    print(f"rep {rep}: plain {steps * N_SENT / dt_plain / 1e6:.1f} M sentences/s ({dt_plain / nb_total * 1e6:.1f} us per batch); clocked {steps * N_SENT / dt / 1e6:.1f} M: per batch "
          f"sync {acc['sync'] * per:.1f} us ({waits / nb_total:.0%} of the syncs waited > 5 us), enqueue {acc['enqueue'] * per:.1f} us, python + rest {(dt - acc['sync'] - acc['enqueue']) * per:.1f} us")

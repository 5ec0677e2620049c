#!/usr/bin/env python
"""Concurrent one-sentence callers (kgpu_debug_concurrent_callers) over and over in ONE process, each run with the cgroup's CPU accounting around it:
python tools/callers_probe.py [threads=128] [calls=200] [runs=12] -- per run: sentences/s, p50 / p99 us, CPU seconds used (cpu.stat usage_usec), the number of
100 ms periods in which the group was throttled and for how long.  Says whether a slow run is the cgroup quota's doing (16 CPUs of the host's 256 hardware
threads on this pool's boxes) and how far below the quota the callers' own CPU use is."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
from kanpyo_amd import Tokenizer, synth
from kanpyo_amd.tokenizer import concurrent_callers, pack_sentences


def cpu_stat():
    try:
        return {k: int(v) for k, v in (line.split() for line in open("/sys/fs/cgroup/cpu.stat"))}
    except (OSError, ValueError):
        return {}


threads = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
sd = synth.build_dict(); tok = Tokenizer(sd.dict)
u, o = pack_sentences(synth.make_corpus(sd, 100000, 1, "cfg2")[:20000])
concurrent_callers(tok, u, o, threads, 20)
print(f"lib {os.environ.get('KGPU_LIB', 'in-tree')}: {threads} threads x {calls} calls, {runs} runs; cpu.max = {open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '?'}")
for r in range(runs):
    if r % 2:
        time.sleep(0.037 * r)  # the runs start at different phases of the quota's 100 ms period
    s0 = cpu_stat()
    concurrent_callers(tok, u, o, threads, 20)  # bench.py's warm-up run directly before the measured one
    s1 = cpu_stat()
    x = concurrent_callers(tok, u, o, threads, calls)
    s2 = cpu_stat()
    d = lambda a, b, k: (b.get(k, 0) - a.get(k, 0))
    print(f"  run {r:2d}: {x['sentences_per_s'] / 1e3:6.0f} k sentences/s  p50 {x['p50_us']:6.0f}  p99 {x['p99_us']:8.0f} us  wall {x['wall_s'] * 1e3:6.1f} ms | "
          f"cpu {d(s1, s2, 'usage_usec') / 1e3:7.1f} ms ({d(s1, s2, 'usage_usec') / max(x['calls'], 1):5.1f} us per call), throttled {d(s1, s2, 'nr_throttled')} periods {d(s1, s2, 'throttled_usec') / 1e3:7.1f} ms"
          f" | warm-up: cpu {d(s0, s1, 'usage_usec') / 1e3:6.1f} ms, throttled {d(s0, s1, 'throttled_usec') / 1e3:6.1f} ms", flush=True)
    if os.environ.get("KGPU_SMALL_TRACE"):
        import ctypes as C
        from kanpyo_amd import _lib
        sc = (C.c_uint64 * 16)(); st = (C.c_uint64 * 8)()
        _lib.lib().kgpu_debug_small_cpu(sc); _lib.lib().kgpu_debug_small_trace(st)
        names = ["joined", "led", "entry+lock", "follower wait", "window", "close+ctx", "assemble", "launch call", "poll", "hand out", "wake", "hipSetDevice"]
        nc = max(x["calls"], 1)
        print(f"           user {d(s1, s2, 'user_usec') / 1e3:.0f} ms, system {d(s1, s2, 'system_usec') / 1e3:.0f} ms of the group; the calling threads' own clocks {x['caller_cpu_s'] * 1e3:.0f} ms = {x['caller_cpu_s'] * 1e6 / nc:.1f} us per call;"
              f" joined {sc[0]}, led {sc[1]} (incl. the warm-up's); per call (us of CPU): " + ", ".join(f"{names[k]} {sc[k] / 1e3 / (sc[0] + sc[1]):.2f}" for k in range(2, 12)), flush=True)
        print(f"           per launch, wall us: prep {st[1] / 1e3 / max(st[0], 1):.1f}, launch call {st[2] / 1e3 / max(st[0], 1):.1f}, poll {st[3] / 1e3 / max(st[0], 1):.1f}, hand out {st[4] / 1e3 / max(st[0], 1):.1f}; {st[5] / max(st[0], 1):.1f} sentences per launch")

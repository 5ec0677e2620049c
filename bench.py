#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: sentences/sec (+ input MiB/s) of
Tokenizer::tokenize on MI355X, synthetic IPADIC-shaped dictionary.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A STEP is one pass of the hot path over one whole 100 000-sentence corpus in
batches of 4096 (24 full batches and the 1 696-sentence tail), inputs resident in
HBM, dense token streams left in HBM.

  N = 1   BASELINE configs[1] (SURVEY 8d cfg 2): the seed-1 corpus, every step.
  N > 1   BASELINE configs[3] (cfg 4): step k is the corpus of seed 100 + k (cycled
          over the corpora generated), sentence i -> GPU i mod N, dictionary
          replicated, no data-path collective; the token records of every step are
          gathered to rank 0 (flat gatherv over xGMI) inside the timed region, so
          `value` is the whole job's rate and `scaling` is "strong" (the work of a
          step does not grow with N).

Rank 0 prints ONE JSON line.  At N = 1 it also carries: the per-launch and per-stage
roofline (HIP events around the kernels; stage split by the runtime's measurement-only
ablation mode), the other single-GPU configs as `extra` lines (cfg 3, cfg 5), call
latencies of the host-buffer entry point, and the CPU baseline (the oracle restatement
timed single-pass on the host cores).

run_job() / Workload are importable: tests/test_dist_cpu.py drives them with world
size 2 over gloo and a CPU engine.
"""
import argparse
import json
import os
import sys
import time

# Four launches side by side are the optimum on MI355X (3: 68.6, 4: 71.5, 5: 58 M sentences/s); HIP's default of 4 hardware
# queues leaves its streams three.  The variable is read by the HIP runtime when it initialises: libkanpyo_gpu.so sets GPU_MAX_HW_QUEUES=8
# itself when it is loaded before the first HIP call (kgpu_api.cpp: kgpu_preinit) -- bench.py therefore loads the library before it touches
# torch.cuda and does NOT set the variable; the line reports the streams the library runs on (config.streams; 4 = the variable took effect).

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
N_SENT = 100_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CHIP_SIMDS, CHIP_CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md)


def algorithmic_bytes(w):
    """Stage A (lattice build) B+16T+C+16N, Stage B (Viterbi) 8E+14N,
    Stage C (backtrace+emit) 28K -- all three run inside the one fused kernel."""
    a = w["B"] + 16 * w["T"] + w["C"] + 16 * w["N"]
    b = 8 * w["E"] + 14 * w["N"]
    c = 28 * w["K"]
    return a, b, c


ROOFLINE_FIRST = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "stage_A_ms", "stage_B_ms", "stage_C_ms", "stage_A_frac", "stage_B_frac",
                  "stage_C_frac", "valu_issue_frac", "valu_cycles_per_wave_op", "peak_measured_read", "frac_of_measured_read", "achieved_at_job_rate", "frac_at_job_rate",
                  "kernel_alone_ms", "frac_alone", "avg_kernel_ms", "launches_in_flight")


def hoist_roofline(r):
    """The driver's record keeps the first ~23 scalar keys of `roofline` and drops nested objects: the north star's per-stage figures (stage B = connection-cost
    gather + Viterbi sweep against the HBM roofline), the VALU issue fraction and the measured streaming read go first, as scalars; the nested forms stay behind them."""
    st = r.get("stages") or {}
    for k, name in (("A", "A_lattice"), ("B", "B_viterbi"), ("C", "C_emit")):
        if name in st:
            r[f"stage_{k}_ms"] = st[name]["ms_per_step"]
            r[f"stage_{k}_frac"] = st[name]["frac"]
    ins = r.get("instruction") or {}
    if "valu_issue_frac" in ins:
        r["valu_issue_frac"] = ins["valu_issue_frac"]
        r["valu_cycles_per_wave_op"] = ins["cycles_per_wave_op"]
    out = {k: r[k] for k in ROOFLINE_FIRST if k in r}
    out.update({k: v for k, v in r.items() if k not in out})
    return out


def c_getenv(name):
    """The C environment (os.environ is Python's start-up snapshot: it does not see the setenv of the library's load-time constructor)."""
    import ctypes

    g = ctypes.CDLL(None).getenv
    g.restype = ctypes.c_char_p
    v = g(name.encode())
    return v.decode() if v else None


def result_rate_guess(rate_1thread, nthreads):
    """Sentences per second to expect from `nthreads` host threads (sizes the all-core leg to about two seconds)."""
    return rate_1thread * max(1.0, 0.5 * nthreads)


def cpu_quota():
    """CPUs this process may actually use at once: the cgroup's CPU quota (cpu.max = "quota period") if one is set, else the affinity
    mask.  The GPU boxes of this pool show 256 hardware threads and a quota of 16."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(round(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cgroup_cpu_stat():
    """The cgroup's CPU accounting (cpu.stat: usage_usec, nr_throttled, throttled_usec ...), {} where there is none."""
    try:
        return {k: int(v) for k, v in (line.split() for line in open("/sys/fs/cgroup/cpu.stat"))}
    except (OSError, ValueError):
        return {}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------ workload

class Workload:
    """The batches one rank owns: for every corpus, the sentences i with i mod world == rank
    (kanpyo_amd.dist.shard_indices) in ascending order, cut into batches of at most `batch`."""

    def __init__(self, corpora, rank=0, world=1, batch=BATCH):
        from kanpyo_amd.dist import shard_indices
        from kanpyo_amd.tokenizer import pack_sentences

        self.rank, self.world, self.batch = rank, world, batch
        self.n_total = [len(c) for c in corpora]
        self.packed = []  # [corpus][b] = (utf8 uint8[], offsets uint64[n+1])
        for c in corpora:
            mine = shard_indices(len(c), rank, world)
            local = [c[i] for i in mine]
            self.packed.append([pack_sentences(local[lo : lo + batch]) for lo in range(0, max(len(local), 1), batch)])

    def n_corpora(self):
        return len(self.packed)

    def nb(self, step):
        return len(self.packed[step % len(self.packed)])

    def sentences(self, step):  # local
        return sum(len(o) - 1 for _, o in self.packed[step % len(self.packed)])

    def bytes_in(self, step):
        return sum(int(o[-1]) for _, o in self.packed[step % len(self.packed)])

    def cap(self):  # tokens <= chars + 1 <= bytes + 1 per sentence: never too small
        return max(int(o[-1]) + len(o) - 1 for p in self.packed for _, o in p) + 8


class GpuEngine:
    """Q device contexts over shared streams; inputs uploaded once, every batch's dense tokens stay in HBM in
    a ring of output buffers (`ring` steps deep: a step's records must survive until its gather is through)."""

    def __init__(self, tok, dev, wl, queue=6, streams=3, ring=1, compact=False):
        """compact: results as 8-byte kgpu_token8 records + the first token's (position, start) per sentence (kgpu_tokenize_device_compact):
        a third of the bytes for the gather; results() then appends the firsts (as int64) behind the counts."""
        import torch

        from kanpyo_amd.device import DeviceContext

        self.torch, self.dev, self.wl, self.Q, self.ring = torch, dev, wl, max(1, queue), ring
        self.inputs = [[(torch.from_numpy(u.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), len(o) - 1, int(o[-1]))
                        for u, o in p] for p in wl.packed]
        self.cap = wl.cap()
        nbmax = max(len(p) for p in wl.packed)
        # token offsets of a step's batches: rows of ONE tensor, so that results() gets the per-sentence counts of the whole
        # step with two tensor ops instead of three per batch (the host side of a gather chunk is what limits N = 8)
        self.off2d = [torch.zeros((nbmax, wl.batch + 1), dtype=torch.int64, device=dev) for _ in range(ring)]
        self.compact = compact
        self.out = [[(torch.empty((self.cap, 2 if compact else 6), dtype=torch.int32, device=dev), self.off2d[r][b],
                      torch.empty(wl.batch, dtype=torch.uint8, device=dev)) for b in range(nbmax)] for r in range(ring)]
        # compact: the firsts of a step's batches, rows of one tensor like the offsets ([batch, sentence, (position, start)])
        self.first3d = [torch.zeros((nbmax, wl.batch, 2), dtype=torch.int32, device=dev) for _ in range(ring)] if compact else None
        # streams = 0: the contexts share the dictionary's own streams (kgpu_ctx_create with a NULL stream: four with GPU_MAX_HW_QUEUES >= 5,
        # else three) -- what a single-GPU caller should do: every further stream in the process competes for the hardware queues (four idle
        # torch streams next to the library's cost a large host call 61 -> 50 M sentences/s, tools/e2e_probe.py PROBE_ENG).  The multi-rank path
        # needs torch streams: it orders them behind the gather's events.
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, min(streams, self.Q)))] if streams > 0 else []
        self.ctxs = [DeviceContext(tok, self.streams[i % len(self.streams)].cuda_stream if self.streams else None) for i in range(self.Q)]
        self.seq, self.occupant, self.where, self.ntok = 0, [None] * self.Q, {}, {}

    def nb(self, step):
        return self.wl.nb(step)

    def enqueue(self, step, b):
        i = self.seq % self.Q
        self.seq += 1
        if self.occupant[i] is not None:
            self.ntok[self.occupant[i]] = self.ctxs[i].sync()
        d_utf8, d_off, n, total = self.inputs[step % len(self.inputs)][b]
        t, o, st = self.out[step % self.ring][b]
        if self.compact:
            self.ctxs[i].tokenize_compact(d_utf8.data_ptr(), d_off.data_ptr(), n, total, t.data_ptr(), self.cap,
                                          self.first3d[step % self.ring][b].data_ptr(), o.data_ptr(), st.data_ptr())
        else:
            self.ctxs[i].tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, total, t.data_ptr(), self.cap, o.data_ptr(), st.data_ptr())
        self.occupant[i] = (step, b)
        self.where[(step, b)] = i

    def _retire(self, key):
        if key not in self.ntok:
            i = self.where[key]
            self.ntok[key] = self.ctxs[i].sync()
            self.occupant[i] = None
        self.where.pop(key, None)
        return self.ntok.pop(key)

    def results(self, step):
        """Waits for the step's batches; -> (token views [k, 6] int32, per-sentence token counts int64), all in HBM."""
        views, nb, total = [], self.nb(step), 0
        for b in range(nb):
            k = self._retire((step, b))
            views.append(self.out[step % self.ring][b][0][:k])
            n = self.inputs[step % len(self.inputs)][b][2]
            assert n == self.wl.batch or b == nb - 1, "only the last batch of a step may be ragged"
            total += n
        if nb == 0:
            return views, self.torch.zeros(0, dtype=self.torch.int64, device=self.dev)
        o = self.off2d[step % self.ring][:nb]
        counts = (o[:, 1:] - o[:, :-1]).reshape(-1)[:total]  # row-major: the full batches, then the ragged one's prefix
        if self.compact:  # [counts (total) | firsts (total, one int64 = (position, start) each)]
            f = self.first3d[step % self.ring][:nb].reshape(-1, 2)[:total].contiguous().view(self.torch.int64).reshape(-1)
            counts = self.torch.cat([counts, f])
        return views, counts

    def after_gather(self):
        """Marks the transfers just waited for (on the RCCL backend work.wait() only makes torch's current stream wait,
        neither the host nor the streams the tokenize kernels run on); order_behind(mark) puts the contexts' streams
        behind it.  The caller does that one chunk LATER, when the ring slot is actually reused: ordering the streams
        behind a mark just recorded stalls every tokenize stream until the copy kernels queued on a full chip are through
        (measured: 57 instead of 68 M sentences/s on the one-rank self-test)."""
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.dev))
        return ev

    def order_behind(self, ev):
        for st in self.streams:
            st.wait_event(ev)

    def drain(self):
        for i, c in enumerate(self.ctxs):
            if self.occupant[i] is not None:
                self.ntok[self.occupant[i]] = c.sync()
                self.occupant[i] = None
        self.where.clear()
        self.ntok.clear()
        self.torch.cuda.synchronize()

    def close(self):
        self.drain()
        for c in self.ctxs:
            c.close()


def run_job(engine, nsteps, gather=None, chunk_steps=1, on_chunk=None):
    """Exactly `nsteps` steps.  With `gather` (a kanpyo_amd.dist.ChunkedGather; every rank passes one): the token
    records of every step travel to the root in chunks of `chunk_steps` steps -- chunk c is posted once chunk
    c + 1 has been enqueued (so it travels while c + 1 is tokenized) and must have left its buffers before chunk
    c + 3 is enqueued (the engine's output ring is three chunks deep).  on_chunk(first_step, result) is called on
    every rank for every finished chunk (result is None off the root)."""
    if gather is None:
        for s in range(nsteps):
            for b in range(engine.nb(s)):
                engine.enqueue(s, b)
        engine.drain()
        return
    posted = []  # first step of every chunk posted, in order; finished ones are consumed from the front
    trace = os.environ.get("BENCH_TRACE_HOST")  # where the host's time goes: results / post / finish / enqueue, ms per chunk on stderr
    acc = {"results": 0.0, "post": 0.0, "finish": 0.0, "enqueue": 0.0}

    def post(c0):
        t0 = time.perf_counter()
        views, counts = [], []
        for s in range(c0, min(c0 + chunk_steps, nsteps)):
            v, c = engine.results(s)
            views += v
            counts.append(c)
        import torch

        t1 = time.perf_counter()
        gather.post_steps(views, torch.cat(counts), copy_own=True)
        posted.append(c0)
        acc["results"] += t1 - t0
        acc["post"] += time.perf_counter() - t1

    def finish_all():
        t0 = time.perf_counter()
        for c0, r in zip(posted, gather.finish()):
            if on_chunk is not None:
                on_chunk(c0, r)
        posted.clear()
        acc["finish"] += time.perf_counter() - t0
        return engine.after_gather()

    starts = list(range(0, nsteps, chunk_steps))
    mark = None  # transfers of the chunks <= k - 3, marked one iteration ago
    for k, c0 in enumerate(starts):
        if mark is not None:
            engine.order_behind(mark)  # chunk k reuses chunk k - 3's ring slot: only behind that chunk's transfers
            mark = None
        if k >= 2:
            mark = finish_all()  # chunks <= k - 2 have left their buffers (host-side on gloo, stream-side on RCCL)
        t0 = time.perf_counter()
        for s in range(c0, min(c0 + chunk_steps, nsteps)):
            for b in range(engine.nb(s)):
                engine.enqueue(s, b)
        acc["enqueue"] += time.perf_counter() - t0
        if k >= 1:
            post(starts[k - 1])
    if starts:
        post(starts[-1])
    finish_all()
    engine.drain()
    if trace:
        print("host ms per chunk:", {k: round(1e3 * v / max(len(starts), 1), 3) for k, v in acc.items()}, file=sys.stderr)


def expand_gathered(tok8_all, cnt2_all, sizes, steps_sentences):
    """Host side (checks, consumers): a gathered chunk of 8-byte records -> (24-byte records [T, 6] int32, counts int64), rank-major.
    cnt2_all holds per rank and step [counts | firsts]; steps_sentences[r] = list of that rank's local sentence counts per step of the chunk."""
    from kanpyo_amd.device import expand_tokens

    toks, cnts, at_t, at_c = [], [], 0, 0
    for r, (nt, nc) in enumerate(sizes):
        seg_t, seg_c = tok8_all[at_t : at_t + nt], cnt2_all[at_c : at_c + nc]
        at_t += nt
        at_c += nc
        t0 = c0 = 0
        for n_s in steps_sentences[r]:
            cnt = seg_c[c0 : c0 + n_s].astype(np.int64)
            first = seg_c[c0 + n_s : c0 + 2 * n_s].astype(np.int64).view(np.uint32).reshape(-1, 2)
            k = int(cnt.sum())
            toff = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
            toks.append(expand_tokens(np.ascontiguousarray(seg_t[t0 : t0 + k]), toff, first).view(np.int32).reshape(-1, 6))
            cnts.append(cnt)
            t0 += k
            c0 += 2 * n_s
        assert t0 == nt and c0 == nc
    return (np.concatenate(toks) if toks else np.zeros((0, 6), np.int32)), (np.concatenate(cnts) if cnts else np.zeros(0, np.int64))


def chunk_steps_for(nb_per_step):
    """Steps per gather chunk: about a dozen batches, so that the host side of a gather (size exchange, one
    grouped send/recv call) stays a small part of the chunk whatever the rank count."""
    return max(1, -(-12 // max(nb_per_step, 1)))


# ------------------------------------------------------------------ extras (N = 1)

def _extras_child(sd, outdir, cfg3_n):
    """Forked before any GPU state exists (a fork is not safe afterwards), but asleep until the parent's timed region is
    over: then it generates the other single-GPU configs' corpora while the parent runs its remaining legs (the
    generator is a pure-Python loop: ~1 minute per million cfg 3 sentences)."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    while not os.path.exists(os.path.join(outdir, "go")):
        time.sleep(0.05)

    # the dense-lattice variant of the dictionary (same record count, N/C ~ 9, more than eight predecessors at about half of the positions) and a
    # cfg 2-shaped corpus over it: SURVEY 8(a) a15's natural density, which the default synthetic shape (N/C = 5.4) does not reach
    sd_dense = synth.build_dict(dense=True)
    sd_dense.dict.save_npz(os.path.join(outdir, "dense_dict.npz"))
    for kind, n, seed, sdx in (("dense", N_SENT, 1, sd_dense), ("cfg5", 1000, 5, sd), ("cfg3", cfg3_n, 2, sd)):
        sents = synth.make_corpus(sdx, n, seed, "cfg2" if kind == "dense" else kind)
        utf8, offs = pack_sentences(sents)
        np.save(os.path.join(outdir, kind + "_utf8.npy"), utf8)
        np.save(os.path.join(outdir, kind + "_offs.npy"), offs)
        np.save(os.path.join(outdir, kind + "_chars.npy"), np.array([sum(map(len, sents))], dtype=np.int64))
        os.rename(os.path.join(outdir, kind + "_chars.npy"), os.path.join(outdir, kind + "_done.npy"))


class PackedWorkload(Workload):
    """A Workload over one already packed corpus (the extras arrive as arrays from the generator process)."""

    def __init__(self, utf8, offs, batch=BATCH):
        self.rank, self.world, self.batch = 0, 1, batch
        n = len(offs) - 1
        self.n_total = [n]
        p = []
        for lo in range(0, max(n, 1), batch):
            hi = min(lo + batch, n)
            p.append((utf8[int(offs[lo]) : int(offs[hi])], (offs[lo : hi + 1] - offs[lo]).astype(np.uint64)))
        self.packed = [p]


def measure_config(tok, dev, wl, n_chars, passes, queue, streams, label, orc=None):
    """One extra config: algorithmic bytes from the device work counters, then `passes` timed passes.  orc (the CPU checker):
    the first batch's records are compared with the oracle's before anything is timed."""
    import torch

    from kanpyo_amd.device import PROFILE_OFF, PROFILE_WORK

    eng = GpuEngine(tok, dev, wl, queue=queue, streams=streams, ring=1)
    plan = eng.ctxs[0].plan()
    bit_exact = None
    if orc is not None:
        eng.enqueue(0, 0)
        k = eng._retire((0, 0))
        t, o, _ = eng.out[0][0]
        u0, o0 = wl.packed[0][0]
        n0 = len(o0) - 1
        exp = orc.tokenize_batch(u0, o0, min(os.cpu_count() or 1, 64))
        bit_exact = bool(k == len(exp.tokens) and np.array_equal(o[: n0 + 1].cpu().numpy().astype(np.uint64), exp.offsets)
                         and np.array_equal(t[:k].cpu().numpy().reshape(-1), exp.tokens.view(np.int32).reshape(-1)))
        eng.drain()
    for c in eng.ctxs:
        c.set_profiling(PROFILE_WORK)
    run_job(eng, 1)
    work = {k: 0 for k in ("sentences", "B", "C", "T", "N", "E", "K")}
    for c in eng.ctxs:
        for k, v in c.work().items():
            work[k] += v
        c.set_profiling(PROFILE_OFF)
        c.profile(reset=True)
    run_job(eng, max(2, -(-eng.Q // max(wl.nb(0), 1))))  # warm: every context has grown its scratch arena, the routing estimate has settled
    # ---- wavefront-slot occupancy: the kernels' own busy time (shader clocks between taking a sentence and its last store, summed over the wavefronts:
    # the profiling instantiation's phase clocks, byte-step counting left out) over the duration of THAT pass x the slots the launch plan keeps resident
    from kanpyo_amd.device import PROFILE_NO_T

    for c in eng.ctxs:
        c.set_profiling(PROFILE_WORK | PROFILE_NO_T)
        c.phase_cycles(reset=True)
    prof_passes = max(1, passes // 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_job(eng, prof_passes)
    dt_prof = (time.perf_counter() - t0) / prof_passes
    busy = 0.0
    for c in eng.ctxs:
        busy += float(sum(c.phase_cycles(reset=True).values())) / prof_passes
        c.work(reset=True)
        c.set_profiling(PROFILE_OFF)
    slots = plan["compute_units"] * max(plan["pool_workgroups_per_cu"] * plan["pool_wavefronts"], plan["window_workgroups_per_cu"])
    for c in eng.ctxs:
        c.profile(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_job(eng, passes)
    dt = (time.perf_counter() - t0) / passes
    # (the profiled pass itself runs far below the product's rate -- every batch's counters are read back with a stream synchronisation -- so the busy clocks
    # per pass are put against the PRODUCT's pass time: sentences/s x busy clocks per sentence / (clock x slots), the review's formula, with the profiling
    # instantiation's clocks, which its own timers inflate by a few per cent)
    slot_occupancy = busy / (dt * CHIP_CLOCK_HZ * max(slots, 1))
    prof = {"batches": 0, "sentences": 0, "deferred": [0] * 4, "redone": [0] * 4, "long_launches": 0, "arena_regrows": 0}
    for c in eng.ctxs:
        p = c.profile(reset=True)
        for k in prof:
            prof[k] = [x + y for x, y in zip(prof[k], p[k])] if isinstance(prof[k], list) else prof[k] + p[k]
    eng.close()
    n = wl.sentences(0)
    a, b, c_ = algorithmic_bytes(work)
    return {
        "workload": label, "sentences": n, "chars_per_sentence": n_chars / max(n, 1), "value": n / dt, "unit": "sentences/s",
        "Mchar_per_s": n_chars / dt / 1e6, "input_MiB_per_s": wl.bytes_in(0) / dt / 2**20, "ms_per_pass": dt * 1e3, "passes": passes,
        "work_per_sentence": {k: work[k] / max(work["sentences"], 1) for k in ("B", "C", "T", "N", "E", "K")},
        "algorithmic_bytes_per_pass": a + b + c_,
        "roofline_at_job_rate": {"achieved": (a + b + c_) / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (a + b + c_) / dt / 1e9 / HBM_PEAK_GBS},
        "slot_occupancy": slot_occupancy,
        "slot_occupancy_what": f"busy shader clocks of the wavefronts per pass ({busy:.4g}: the kernels' own phase clocks, profiling instantiation) / "
                               f"(a timed pass's {dt * 1e3:.3f} ms x {CHIP_CLOCK_HZ / 1e9:.1f} GHz x {slots} resident wavefront slots); the clocks are the profiling instantiation's "
                               f"(a few per cent above the product's), the profiled pass itself took {dt_prof * 1e3:.3f} ms",
        "routing": prof,
        "batch": wl.batch, "batches_per_pass": wl.nb(0), "batches_in_flight": eng.Q,
        "first_batch_bit_exact_vs_oracle": bit_exact,
        "launch_plan": plan,
        "lds_bytes_per_workgroup": {"pool_kernel": plan["pool_lds_bytes"], "windowed_kernel": plan["window_lds_bytes"]},
        "resident_workgroups_per_cu": {"pool_kernel": plan["pool_workgroups_per_cu"], "windowed_kernel": plan["window_workgroups_per_cu"]},
        "resident_wavefronts_per_cu": {"pool_kernel": plan["pool_workgroups_per_cu"] * plan["pool_wavefronts"], "windowed_kernel": plan["window_workgroups_per_cu"]},
    }


# ------------------------------------------------------------------ one process, several devices (the C ABI's own multi-device entry)

def run_single_process(args):
    """bench.py --gpus N --single-process [--devices 0,1,...]: cfg 4 through kgpu_multi_* -- ONE process, no torch.distributed: sentence i -> entry
    i mod N, every entry's shard resident on its device, the compaction kernels store the 8-byte records straight into the ROOT device's memory over
    xGMI (peer access): the stores are the gather.  torch is used for device memory only.  The same entry may name one device several times
    (--devices 0,0: the path's self-test on one GPU).  Prints the same line fields as the torch.distributed path."""
    import ctypes as C

    import torch

    from kanpyo_amd import Tokenizer, _lib, synth
    from kanpyo_amd.dist import reassemble
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences

    L = _lib.lib()  # before the first HIP call: the library asks for its hardware queues itself
    G, K, W, Q = args.gpus, args.steps, args.warmup, args.queue
    ndev = torch.cuda.device_count()
    devices = [int(x) for x in args.devices.split(",")] if args.devices else [g % max(ndev, 1) for g in range(G)]
    assert len(devices) == G and all(0 <= d < ndev for d in devices), (devices, ndev)
    sd = synth.build_dict()
    ncorp = max(1, min(args.corpora if args.corpora > 0 else 100, 100, max(K, 1)))
    corpora = [synth.make_corpus(sd, N_SENT, seed=100 + k, kind="cfg2") for k in range(ncorp)]
    toks = {}
    for d in devices:  # one dictionary handle per distinct device
        if d not in toks:
            toks[d] = Tokenizer(sd.dict, device=d)
    handles = (C.c_void_p * G)(*[toks[d].handle for d in devices])
    mh = C.c_void_p()
    _lib.check(L.kgpu_multi_create(handles, G, Q, C.byref(mh)))
    root = torch.device("cuda", devices[0])
    # inputs: per corpus, per entry, per batch -- resident on the entry's device
    wls = [Workload(corpora, g, G) for g in range(G)]
    nb = max(w.nb(0) for w in wls)
    cap = max(w.cap() for w in wls)
    inputs = []  # [corpus][b][g] = (utf8, offsets, n, total)
    for ci in range(ncorp):
        per_b = []
        for b in range(nb):
            row = []
            for g in range(G):
                dev = torch.device("cuda", devices[g])
                if b < len(wls[g].packed[ci]):
                    u, o = wls[g].packed[ci][b]
                else:
                    u, o = np.zeros(0, np.uint8), np.zeros(1, np.uint64)
                row.append((torch.from_numpy(np.ascontiguousarray(u)).to(dev) if u.size else torch.zeros(16, dtype=torch.uint8, device=dev),
                            torch.from_numpy(o.astype(np.int64)).to(dev), len(o) - 1, int(o[-1])))
            per_b.append(row)
        inputs.append(per_b)
    outs = [[dict(t8=torch.empty((cap, 2), dtype=torch.int32, device=root), first=torch.empty(2 * BATCH + 2, dtype=torch.int32, device=root),
                  toff=torch.empty(BATCH + 1, dtype=torch.int64, device=root), st=torch.empty(BATCH + 16, dtype=torch.uint8, device=root)) for _ in range(G)]
            for _ in range(Q)]
    ptrs = lambda rows, k: (C.c_void_p * G)(*[r[k].data_ptr() for r in rows])
    u64s = lambda vals: (C.c_uint64 * G)(*vals)
    pending = [None] * Q
    got = (C.c_uint64 * G)()
    tokens_total = [0]

    def retire(slot):
        if pending[slot] is not None:
            _lib.check(L.kgpu_multi_sync(mh, slot, got))
            tokens_total[0] += sum(int(x) for x in got)
            pending[slot] = None

    def job(nsteps, keep=None):
        k = 0
        for s_ in range(nsteps):
            for b in range(nb):
                slot = k % Q
                retire(slot)
                rows = inputs[s_ % ncorp][b]
                o = outs[slot]
                _lib.check(L.kgpu_multi_tokenize_device(
                    mh, slot, (C.c_void_p * G)(*[r[0].data_ptr() for r in rows]), (C.c_void_p * G)(*[r[1].data_ptr() for r in rows]),
                    u64s([r[2] for r in rows]), u64s([r[3] for r in rows]),
                    (C.c_void_p * G)(*[x["t8"].data_ptr() for x in o]), u64s([cap] * G), (C.c_void_p * G)(*[x["first"].data_ptr() for x in o]),
                    (C.c_void_p * G)(*[x["toff"].data_ptr() for x in o]), (C.c_void_p * G)(*[x["st"].data_ptr() for x in o])))
                pending[slot] = (s_, b)
                if keep is not None:  # the untimed check wants every batch's records: retire at once and copy them out
                    retire(slot)
                    keep.append([(x["t8"][: int(got[g])].cpu().numpy().copy(), x["first"][: 2 * rows[g][2]].cpu().numpy().copy(),
                                  x["toff"][: rows[g][2] + 1].cpu().numpy().copy()) for g, x in enumerate(o)])
                k += 1
        for slot in range(Q):
            retire(slot)

    # ---- untimed: one step gathered, expanded and reassembled == the same corpus tokenized on the root device alone
    kept = []
    job(1, keep=kept)
    toks24 = [[] for _ in range(G)]
    cnts = [[] for _ in range(G)]
    for batch_rows in kept:
        for g, (t8, first, toff) in enumerate(batch_rows):
            n_g = len(toff) - 1
            out = np.empty(len(t8), dtype=TOKEN_DTYPE)
            toff_u, t8_c, first_u = toff.astype(np.uint64), np.ascontiguousarray(t8), np.ascontiguousarray(first.astype(np.uint32))  # (named: they must outlive the call)
            L.kgpu_expand_tokens(t8_c.ctypes.data, toff_u.ctypes.data, first_u.ctypes.data, n_g, out.ctypes.data)
            toks24[g].append(out.view(np.int32).reshape(-1, 6))
            cnts[g].append(np.diff(toff).astype(np.int64))
    g_tok, g_off = reassemble(np.concatenate([np.concatenate(x) if x else np.zeros((0, 6), np.int32) for x in toks24]),
                              np.concatenate([np.concatenate(x) if x else np.zeros(0, np.int64) for x in cnts]), len(corpora[0]), G)
    u0, o0 = pack_sentences(corpora[0])
    one_t, one_off, _ = toks[devices[0]].tokenize_packed(u0, o0)
    gather_check = bool(np.array_equal(g_off.astype(np.uint64), one_off) and np.array_equal(g_tok.reshape(-1), one_t.view(np.int32).reshape(-1)))
    if not gather_check:
        same_off = np.array_equal(g_off.astype(np.uint64), one_off)
        bad = np.nonzero(np.diff(g_off.astype(np.int64)) != np.diff(one_off.astype(np.int64)))[0]
        print(f"single-process check: offsets equal {same_off}; {len(g_off)} vs {len(one_off)} offsets, {g_tok.shape} vs {one_t.shape} tokens; first differing sentences {bad[:8]}", file=sys.stderr)
    assert gather_check, "gathered + reassembled token stream differs from the single-device stream"

    def sync_all():
        for d in set(devices):
            torch.cuda.synchronize(d)

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        job(4)
    if W > 0:
        job(W)
    tokens_total[0] = 0
    sync_all()
    t0 = time.perf_counter()
    job(K)
    sync_all()
    elapsed = time.perf_counter() - t0
    sentences = sum(len(corpora[s_ % ncorp]) for s_ in range(K))
    distinct = len(set(devices))
    result = {
        "metric": "sentences/sec", "value": sentences / elapsed, "unit": "sentences/s", "n_gpus": G, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3] (cfg 4): 100k-sentence corpora of seeds 100..{99 + ncorp} cycled, sentence i -> entry i mod {G} of ONE process "
                               f"(kgpu_multi_*: devices {devices}), 8-byte records stored into device {devices[0]}'s memory by the shards' compaction kernels; "
                               "synthetic IPADIC-shaped dictionary (392k records); batch=4096 per entry; inputs resident in HBM",
                   "batch": BATCH, "sentences_per_step": N_SENT, "batches_per_step_per_gpu": nb, "batches_in_flight": Q, "devices": devices,
                   "distinct_devices": distinct, "launcher": "single process (C ABI kgpu_multi_create / kgpu_multi_tokenize_device / kgpu_multi_sync), no torch.distributed",
                   "sharding": f"sentence i -> entry i mod {G}, dictionary replicated per device, no data-path collective"},
        "sentences_total": sentences,
        "gather": {"tokens": tokens_total[0], "sentences": sentences, "complete": True, "reassembled_step_equals_one_gpu": gather_check, "record_bytes": 8,
                   "records": "kgpu_token8 (8 bytes) + the first token's (position, start) per sentence, written by every shard's compaction kernel into the root "
                              "device's memory (peer stores over xGMI when the entries are distinct devices); kgpu_expand_tokens restores the 24-byte records where they are consumed",
                   "root_ingest_GB_per_s": tokens_total[0] * 8 * (distinct - 1) / max(distinct, 1) / elapsed / 1e9},
        "per_rank": [{"rank": g, "device": devices[g], "sentences": int(sum(wls[g].sentences(s_) for s_ in range(K)))} for g in range(G)],
        "corpora": {"distinct": ncorp, "seeds": f"100..{99 + ncorp}", "sentences_each": N_SENT},
    }
    L.kgpu_multi_destroy(mh)
    print(json.dumps(result), flush=True)


# ------------------------------------------------------------------ main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queue", type=int, default=8, help="batches in flight (one context each)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the contexts share round-robin (one hardware queue each with "
                    "GPU_MAX_HW_QUEUES=8; a stream that has to share a queue unbalances them)")
    ap.add_argument("--corpora", type=int, default=0, help="N>1: distinct cfg 4 corpora (seeds 100..) generated and cycled; 0 = one per step, "
                    "at most 100 (100 = all of cfg 4: 10 M sentences)")
    ap.add_argument("--no-one-gpu-leg", action="store_true", help="N>1: skip the untimed-region leg in which rank 0 runs the same steps alone (speedup_vs_1gpu)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5, help="untimed: the same steps for this long before the W warmup steps "
                    "(the first process on a freshly started box measures ~4 %% low for its first second: clocks / page tables still settling)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="lower bound of CPU-baseline work")
    ap.add_argument("--cfg3-sentences", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg 3 / cfg 5 / latency / stage legs")
    ap.add_argument("--gather-records", type=int, default=8, choices=(8, 24), help="N>1: bytes per token record on the wire: 8 = kgpu_token8 (+ the first token's "
                    "position / start per sentence; kgpu_expand_tokens restores the 24-byte records on the consumer's side), 24 = kgpu_token")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (self-test)")
    ap.add_argument("--single-process", action="store_true", help="--gpus N in ONE process through the C ABI's multi-device entry (kgpu_multi_*), no torch.distributed")
    ap.add_argument("--devices", default="", help="--single-process: the device of every entry, e.g. 0,0 (default: entry g -> device g)")
    args = ap.parse_args()
    if args.single_process:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            import io
            import contextlib

            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                run_single_process(args)
        finally:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
        print(buf.getvalue().strip().split("\n")[-1], flush=True)
        return

    # stdout carries exactly one JSON line: everything else that libraries print there (RCCL's
    # banner at communicator creation, ...) is routed to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    multi = world > 1 or args.force_dist
    K, W = args.steps, args.warmup

    import torch  # before libkanpyo_gpu.so (build_dict loads it): torch bundles its own libamdhip64 and must win
    import torch.distributed as dist

    from kanpyo_amd import _lib, synth

    _lib.lib()  # before any HIP call of this process: the library asks for the hardware queues it needs (see the top of this file)
    sd = synth.build_dict()
    extras_dir, extras_proc = None, None
    if world == 1 and not args.no_extras:  # before the HIP runtime is initialised in this process: fork is safe
        import multiprocessing as mp
        import tempfile

        extras_dir = tempfile.mkdtemp(prefix="kanpyo_bench_")
        extras_proc = mp.get_context("fork").Process(target=_extras_child, args=(sd, extras_dir, args.cfg3_sentences), daemon=True)
        extras_proc.start()

    assert torch.cuda.is_available(), "bench.py needs an MI355X: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from kanpyo_amd import Tokenizer
    from kanpyo_amd._lib import kernel_source_hash
    from kanpyo_amd.device import PROFILE_EVENTS, PROFILE_OFF, PROFILE_SAMPLED, PROFILE_WORK, STAGE_ALL, STAGE_LATTICE, STAGE_VITERBI, DeviceContext
    from kanpyo_amd.dist import ChunkedGather, reassemble
    from kanpyo_amd.tokenizer import pack_sentences

    # ---- workload
    if world == 1:
        corpora = [synth.make_corpus(sd, N_SENT, seed=1, kind="cfg2")]
        label = "BASELINE configs[1] (cfg 2): 100k synthetic ~40-char sentences (seed 1)"
    else:
        ncorp = max(1, min(args.corpora if args.corpora > 0 else 100, 100, max(K, 1)))  # distinct seeds for every timed step (SURVEY 8d cfg 4)
        corpora = [synth.make_corpus(sd, N_SENT, seed=100 + k, kind="cfg2") for k in range(ncorp)]
        label = (f"BASELINE configs[3] (cfg 4): 100k-sentence corpora of seeds 100..{99 + ncorp} cycled, sentence i -> GPU i mod {world}, "
                 "token records gathered to rank 0 over xGMI")
    wl = Workload(corpora, rank if world > 1 else 0, world)
    tok = Tokenizer(sd.dict, device=local_rank)
    cs = chunk_steps_for(wl.nb(0))
    compact = multi and args.gather_records == 8
    eng = GpuEngine(tok, dev, wl, queue=args.queue, streams=args.streams if multi else 0, ring=3 * cs if multi else 1, compact=compact)
    Q = eng.Q

    # ---- untimed: device-side work counters of every distinct batch of corpus 0 (algorithmic bytes)
    work = {k: 0 for k in ("sentences", "B", "C", "T", "N", "E", "K")}
    for c in eng.ctxs:
        c.set_profiling(PROFILE_WORK)
    run_job(eng, 1)
    for c in eng.ctxs:
        for k, v in c.work().items():
            work[k] += v
        c.set_profiling(PROFILE_OFF)
    sample_tokens = None
    if rank == 0 and world == 1:  # keep batch 0's GPU result for the bit-exact check in the cpu_baseline leg
        eng.enqueue(0, 0)
        k = eng._retire((0, 0))
        t, o, _ = eng.out[0][0]
        sample_tokens = (t[:k].cpu().numpy().copy(), o.cpu().numpy().copy())
        eng.drain()

    # ---- multi-rank: the gather, and (untimed) the check of one gathered + reassembled step against one GPU
    size_pg = dist.new_group(backend="gloo") if multi else None  # CPU-side size exchange of the gather
    gathered = {"tokens": 0, "sentences": 0, "chunks": 0}

    def count_chunk(c0, r):
        if r is not None:
            gathered["tokens"] += int(r[0].shape[0])
            gathered["sentences"] += int(r[1].shape[0]) // (2 if compact else 1)  # compact: [counts | firsts] per rank and step
            gathered["chunks"] += 1

    def job(nsteps, on_chunk=count_chunk):
        if not multi:
            run_job(eng, nsteps)
        else:
            run_job(eng, nsteps, ChunkedGather(dst=0, size_group=size_pg), cs, on_chunk)

    gather_check = None
    if multi:
        got = {}
        job(1, lambda c0, r: got.update(r=r))
        if rank == 0:
            tok_all, cnt_all, sizes_all = got["r"][:3]
            tok_np, cnt_np = tok_all.cpu().numpy(), cnt_all.cpu().numpy()
            if compact:  # the check wants the 24-byte records: expand the gathered chunk on the host (one step per rank here)
                n0 = len(corpora[0])
                tok_np, cnt_np = expand_gathered(tok_np, cnt_np, sizes_all, [[(n0 - r + world - 1) // world] for r in range(world)])
            g_tok, g_off = reassemble(tok_np, cnt_np, len(corpora[0]), world)
            full = GpuEngine(tok, dev, Workload(corpora[:1], 0, 1), queue=2, streams=0, ring=1)
            for b in range(full.nb(0)):
                full.enqueue(0, b)
            fv, fc = full.results(0)
            f_tok = torch.cat(fv).cpu().numpy()
            f_off = np.concatenate([[0], np.cumsum(fc.cpu().numpy())])
            gather_check = bool(np.array_equal(g_off, f_off) and np.array_equal(g_tok, f_tok))
            full.close()
            del full
        if rank == 0 and world > 1:  # allocator blocks of the chunk sizes: no hipMalloc inside the timed region
            warm = [torch.empty((cs * N_SENT * 40, 2 if compact else 6), dtype=torch.int32, device=dev) for _ in range(3)]
            del warm
        torch.cuda.synchronize()

    # ---- N > 1: the same steps on ONE GPU (rank 0 alone, the others wait): the denominator of speedup_vs_1gpu
    one_gpu = None
    if world > 1 and not args.no_one_gpu_leg:
        if rank == 0:
            eng1 = GpuEngine(tok, dev, Workload(corpora, 0, 1), queue=args.queue, streams=0, ring=1)
            run_job(eng1, max(2, min(W, 5)))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_job(eng1, K)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            n1 = sum(len(corpora[s_ % len(corpora)]) for s_ in range(K))
            one_gpu = {"value": n1 / dt1, "unit": "sentences/s", "ms_per_step": dt1 / K * 1e3,
                       "what": f"the same {K} steps (same corpora, unsharded) on rank 0's GPU alone, no gather, before the timed region"}
            eng1.close()
            del eng1
        dist.barrier()

    # ---- warmup (also brings up the RCCL channels of the gather)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:  # untimed, every rank the same number of rounds
        if not multi:
            job(20)
        else:
            job(4)
            flag = torch.tensor([1.0 if time.perf_counter() - t_pre < args.prewarm_seconds else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) == 0.0:
                break
    if W > 0:
        job(W)
    for c in eng.ctxs:
        if not os.environ.get("BENCH_NO_EVENTS"):
            c.set_profiling(PROFILE_EVENTS | PROFILE_SAMPLED)  # HIP events around every 4th launch chain
        c.profile(reset=True)
    for k in gathered:
        gathered[k] = 0

    # ---- timed region: exactly K steps
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job(K)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0  # this rank's own steps (and, on the root, the gathers it waited for)
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([float(sum(wl.sentences(s_) for s_ in range(K))), local_elapsed], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "sentences": int(x[0].item()), "seconds": float(x[1].item()),
                     "sentences_per_s": float(x[0].item()) / max(float(x[1].item()), 1e-9)} for r, x in enumerate(allr)]

    prof = {"launches": 0, "tokenize_ms": 0.0, "first_ms": 0.0, "aux_ms": 0.0, "batches": 0, "sentences": 0, "deferred": [0] * 4, "redone": [0] * 4,
            "long_launches": 0, "arena_regrows": 0}
    for c in eng.ctxs:
        p = c.profile(reset=True)
        for k in prof:
            prof[k] = [x + y for x, y in zip(prof[k], p[k])] if isinstance(prof[k], list) else prof[k] + p[k]
        c.set_profiling(PROFILE_OFF)

    if rank != 0:
        dist.destroy_process_group()
        return

    sentences = sum(len(corpora[s % len(corpora)]) for s in range(K))  # whole job: all ranks' shards
    per_corpus_bytes = [sum(len(x.encode("utf-8")) for x in c) for c in corpora]
    bytes_in = sum(per_corpus_bytes[s % len(corpora)] for s in range(K))
    a, b, c_ = algorithmic_bytes(work)  # of this rank's shard of corpus 0 (world 1: the whole corpus)
    n_work = max(work["sentences"], 1)
    full_batches = [i for i in range(wl.nb(0)) if len(wl.packed[0][i][1]) - 1 == BATCH]
    per_sentence_bytes = (a + b + c_) / n_work
    per_launch_bytes = per_sentence_bytes * BATCH
    avg_kernel_s = prof["first_ms"] / max(prof["launches"], 1) / 1e3    # the dominant kernel's own launches (what rocprofv3 reports for it)
    avg_chain_s = prof["tokenize_ms"] / max(prof["launches"], 1) / 1e3  # ... plus the small launches behind it and their wait for a slot
    # every 4th launch chain is timed, tail batches included: scale the bytes to the average timed launch
    avg_launch_sentences = wl.sentences(0) / wl.nb(0)
    achieved = per_sentence_bytes * avg_launch_sentences / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    job_rate_bytes = per_sentence_bytes * sentences / elapsed / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from a separate rocprofv3 --pmc pass
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = None

    result = {
        "metric": "sentences/sec", "value": sentences / elapsed, "unit": "sentences/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {
            "workload": label + "; synthetic IPADIC-shaped dictionary (392k records, 1316x1316 i16 matrix, 11 categories, 40 unk rows); "
                        "batch=4096 (24 full batches + the 1696-sentence tail per 100k sentences at N=1); one step = one whole corpus; "
                        "inputs resident in HBM, dense tokens left in HBM",
            "batch": BATCH, "sentences_per_step": N_SENT, "batches_per_step_per_gpu": wl.nb(0), "batches_in_flight": Q,
            "streams": eng.ctxs[0].plan()["streams"], "long_streams": eng.ctxs[0].plan()["long_streams"], "GPU_MAX_HW_QUEUES": c_getenv("GPU_MAX_HW_QUEUES"),
            "sharding": "sentence i -> GPU i mod N, dictionary replicated, one gatherv of token records to rank 0 per chunk of "
                        f"{cs} step(s)" if multi else "single GPU",
        },
        "input_MiB_per_s": bytes_in / elapsed / 2**20,
        "work_per_sentence": {k: work[k] / n_work for k in ("B", "C", "T", "N", "E", "K")},
        "routing": {k: prof[k] for k in ("batches", "sentences", "deferred", "redone", "long_launches", "arena_regrows")},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
            "traffic_stale": (traffic.get("kernel_src_sha16") != kernel_source_hash()) if traffic else None,  # counters measured on other kernel sources than these
            "traffic_source": (traffic.get("source", "profiles/pmc_traffic.json") + " -- separate rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, "
                               "per launch), NOT measured inside this run") if traffic else None,
            "kernel": "k_tokenize_pool (fused lattice build + Viterbi + backtrace, LDS page pool)",
            "algorithmic_bytes_per_sentence": per_sentence_bytes,
            "algorithmic_bytes_per_launch": per_launch_bytes,
            "stage_bytes_per_launch": {"A_lattice": a / n_work * BATCH, "B_viterbi": b / n_work * BATCH, "C_emit": c_ / n_work * BATCH},
            "avg_kernel_ms": avg_kernel_s * 1e3, "launches_timed": prof["launches"],
            "avg_kernel_what": f"HIP events on the ctx stream around the k_tokenize_pool launch of every 4th batch, {Q} batches in flight on the "
                               "chip, four of them running (a launch therefore lasts several times its share of the chip's work; see "
                               "kernel_alone_ms and frac_at_job_rate); profiles/r05_kernel_stats.csv / r05_pool_dispatches.txt hold rocprofv3's durations of the same kernel for the same command",
            "avg_launch_chain_ms": avg_chain_s * 1e3,
            "aux_kernels_avg_ms": prof["aux_ms"] / max(prof["launches"], 1),
            "launches_in_flight": Q,
            "achieved_at_job_rate": job_rate_bytes, "frac_at_job_rate": job_rate_bytes / HBM_PEAK_GBS,
        },
    }
    wps = result["work_per_sentence"]
    # what a sentence holds of its pool in the sweep phase (kgpu_pool.hip's own carve: text, 26 B per character, 8 B per bucket entry, 12 B per node, the block's pair table)
    result["roofline"]["lds_bytes_per_sentence"] = (wps["B"] + 4) + 26 * (wps["C"] + 2) + 8 * (wps["N"] + 2) + 12 * (wps["N"] + 1) + 1024
    result["sentences_total"] = sentences
    if multi:
        result["gather"] = {"chunks": gathered["chunks"], "tokens": gathered["tokens"], "sentences": gathered["sentences"],
                            "complete": gathered["sentences"] == sentences, "reassembled_step_equals_one_gpu": gather_check,
                            "chunk_steps": cs, "record_bytes": 8 if compact else 24,
                            "records": ("kgpu_token8 (8 bytes) + the first token's (position, start) per sentence; the root holds them as gathered, "
                                        "kgpu_expand_tokens restores the 24-byte kgpu_token records on the consumer's side (not in the timed region)") if compact
                                       else "kgpu_token (24 bytes)",
                            "root_ingest_GB_per_s": gathered["tokens"] * (8 if compact else 24) * (world - 1) / max(world, 1) / elapsed / 1e9,
                            "root_ingest_what": "token records arriving at rank 0 from the other ranks over xGMI (its own share, 1/N of the "
                                                "stream, is a local copy), averaged over the timed region"}
        result["per_rank"] = per_rank
        result["corpora"] = {"distinct": len(corpora), "seeds": f"100..{99 + len(corpora)}", "sentences_each": N_SENT}
        if one_gpu is not None:
            result["one_gpu_leg"] = one_gpu
            result["speedup_vs_1gpu"] = result["value"] / one_gpu["value"]
        assert gathered["sentences"] == sentences, (gathered, sentences)
        assert gather_check, "gathered + reassembled token stream differs from the single-GPU stream"

    if world == 1:
        # ---- the dominant kernel alone on the chip: one batch at a time, HIP events around every launch
        c0 = eng.ctxs[0]
        c0.set_profiling(PROFILE_EVENTS)
        for rep in range(2):
            c0.profile(reset=True)
            for bi in full_batches:
                d_utf8, d_off, n, total = eng.inputs[0][bi]
                t, o, st = eng.out[0][bi]
                c0.tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, total, t.data_ptr(), eng.cap, o.data_ptr(), st.data_ptr())
                c0.sync()
        p = c0.profile(reset=True)
        c0.set_profiling(PROFILE_OFF)
        alone_ms = p["first_ms"] / max(p["launches"], 1)
        result["roofline"]["kernel_alone_ms"] = alone_ms
        result["roofline"]["launch_chain_alone_ms"] = p["tokenize_ms"] / max(p["launches"], 1)
        result["roofline"]["achieved_alone"] = per_launch_bytes / (alone_ms * 1e-3) / 1e9 if alone_ms > 0 else None
        result["roofline"]["frac_alone"] = per_launch_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if alone_ms > 0 else None

    if world == 1 and not args.no_extras:
        # ---- per-stage roofline: the same pipeline with every sentence stopped after a stage (measurement-only mode of
        # the runtime); a stage's time is the difference of consecutive stop levels at full occupancy
        # (five repetitions of three steps per level, the levels interleaved, the FASTEST repetition counts: a stopped chain is a 40 us kernel per
        # batch, so a level's time is easily the host's launch rate or one scheduling hiccup instead of the GPU's -- a single 3 ms sample once
        # made stage A 0.84 ms and stage B "141 % of the HBM peak")
        stage_runs = {"A": [], "AB": [], "ABC": []}
        for rep in range(5):
            for name, stop in (("A", STAGE_LATTICE), ("AB", STAGE_VITERBI), ("ABC", STAGE_ALL)):
                for c in eng.ctxs:
                    c.set_ablation(stop)
                run_job(eng, 1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run_job(eng, 3)
                stage_runs[name].append((time.perf_counter() - t1) / 3 * 1e3)
        stage_ms = {k: min(v) for k, v in stage_runs.items()}
        for c in eng.ctxs:
            c.set_ablation(STAGE_ALL)
        sb = {"A_lattice": a, "B_viterbi": b, "C_emit": c_}  # bytes per step (whole corpus)
        sm = {"A_lattice": stage_ms["A"], "B_viterbi": stage_ms["AB"] - stage_ms["A"], "C_emit": stage_ms["ABC"] - stage_ms["AB"]}
        def stage_line(k):
            ok = sm[k] > 0 and sb[k] / (sm[k] * 1e-3) / 1e9 <= HBM_PEAK_GBS   # a difference of two timings can come out at or below zero: then it says nothing
            return {"bytes_per_step": sb[k], "ms_per_step": sm[k],
                    "achieved": sb[k] / (sm[k] * 1e-3) / 1e9 if ok else None,
                    "frac": sb[k] / (sm[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if ok else None}
        result["roofline"]["stages"] = {k: stage_line(k) for k in sb}
        result["roofline"]["stages"]["level_ms_per_step_runs"] = {k: [round(x, 4) for x in v] for k, v in stage_runs.items()}
        result["roofline"]["stages"]["how"] = ("kgpu_ctx_set_ablation: steps timed with every sentence stopped after the lattice build / after the sweep / "
                                                "not at all, full pipeline; five interleaved repetitions of three steps per level, the fastest counts; "
                                                "stage time = difference of consecutive levels (B_viterbi = connection-cost gather + sweep)")
        # ---- instruction roofline: the ceiling this kernel is actually near.  Instruction counts per sentence come from a separate
        # rocprofv3 --pmc pass (profiles/); the rate is this run's.
        ipath = os.path.join(ROOT, "profiles", "pmc_instructions.json")
        if os.path.exists(ipath):
            try:
                ins = json.load(open(ipath))
                valu, salu = ins["valu_per_sentence"], ins["salu_per_sentence"]
                rate = result["value"]
                result["roofline"]["instruction"] = {
                    "valu_per_sentence": valu, "salu_per_sentence": salu, "source": ins.get("source", "profiles/pmc_instructions.json"),
                    "stale": ins.get("kernel_src_sha16") != kernel_source_hash(),
                    # tools/ubench/valu.hip (profiles/experiments/r05_valu_issue_rate.txt): a wave64 op occupies its SIMD for 2 cycles only if it is a plain two-operand
                    # VOP2 add / and / move; DPP forms, VOP3 (v_lshl_add, v_mad, v_add3), v_min, v_cndmask, v_cmp + v_cndmask take 4.  About half of the pool kernel's
                    # VALU instructions are of the first kind (static mix): 3 cycles per op on average, bracketed by the two bounds.
                    "cycles_per_wave_op": 3.0,
                    "valu_issue_frac": valu * 3.0 * rate / (CHIP_SIMDS * CHIP_CLOCK_HZ),
                    "valu_issue_frac_if_all_2_cycle_ops": valu * 2.0 * rate / (CHIP_SIMDS * CHIP_CLOCK_HZ),
                    "valu_issue_frac_if_all_4_cycle_ops": valu * 4.0 * rate / (CHIP_SIMDS * CHIP_CLOCK_HZ),
                    "what": "wave-VALU-instructions per sentence x cycles per op x sentences/s / (1024 SIMDs x 2.4 GHz); SQ_ACTIVE_INST_VALU counts one quad-cycle per "
                            "instruction whatever it costs and is not a busy time"}
            except Exception as e:
                print(f"instruction roofline skipped: {e}", file=sys.stderr)

    # ---- the box's own streaming-read bandwidth (a 4 GiB int64 reduction, best of 5): second denominator of the roofline
    if world == 1:
        try:
            x = torch.empty(1 << 29, dtype=torch.int64, device=dev).fill_(1)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 0.0
            for _ in range(6):
                ev0.record(); x.sum(); ev1.record(); ev1.synchronize()
                best = max(best, x.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9)
            del x
            result["roofline"]["peak_measured_read"] = best
            result["roofline"]["frac_of_measured_read"] = achieved / best
        except Exception as e:  # never let the auxiliary measurement break the bench line
            result["roofline"]["peak_measured_read"] = None
            print(f"streaming-read measurement skipped: {e}", file=sys.stderr)

    # ---- host-buffer entry point (H2D + kernels + D2H per call): PCIe-inclusive rates and call latencies, never `value`
    if world == 1:
        from kanpyo_amd.tokenizer import TOKEN_DTYPE, pinned_empty
        utf8_0, offs_0 = wl.packed[0][0]
        cap = eng.cap
        h_out = (np.empty(cap, dtype=TOKEN_DTYPE), np.empty(BATCH + 1, dtype=np.uint64), np.empty(BATCH, dtype=np.uint8))
        tok.tokenize_packed(utf8_0, offs_0, out=h_out)  # untimed: the pool ctx allocates its scratch, pages get touched
        per_call = []
        for i in range(min(wl.nb(0), 12) * 2):
            u, o = wl.packed[0][i % min(wl.nb(0), 12)]
            t1 = time.perf_counter()
            tok.tokenize_packed(u, o, out=h_out)
            per_call.append((time.perf_counter() - t1) / (len(o) - 1))
        per_call.sort()
        result["pcie_inclusive"] = {"value": 1.0 / per_call[len(per_call) // 2], "unit": "sentences/s",
                                    "what": "kgpu_tokenize_batch: pageable host buffers in, dense tokens out, one 4096-sentence call at a time (median of 24 calls over "
                                            "12 different batches; a call that has to allocate a context's scratch is a millisecond-scale outlier)",
                                    "slowest_call_sentences_per_s": 1.0 / per_call[-1]}
        lat = {}
        for n_call in (1, 64, 4096):  # the reference's call shape is n = 1: Tokenizer::tokenize(&str), once per CLI line
            o = offs_0[: n_call + 1].copy()
            u = utf8_0[: int(o[-1])]
            for _ in range(20):
                tok.tokenize_packed(u, o, out=h_out)
            ts = []
            for _ in range(200 if n_call < 4096 else 50):
                t1 = time.perf_counter()
                tok.tokenize_packed(u, o, out=h_out)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            lat[f"n{n_call}"] = {"median_us": ts[len(ts) // 2] * 1e6, "p10_us": ts[len(ts) // 10] * 1e6,
                                 "sentences_per_s_at_median": n_call / ts[len(ts) // 2]}
        result["pcie_inclusive"]["call_latency"] = lat
        result["pcie_inclusive"]["call_latency_what"] = ("kgpu_tokenize_batch through the ctypes mirror (Tokenizer.tokenize_packed, caller-owned "
                                                         "result arrays), host buffers in and out, wall time per call; n <= 128 takes the single-launch "
                                                         "path (pinned in/out, the kernel compacts and publishes itself), of which ~40 us are the one "
                                                         "sentence's own dependent chain on one wavefront")
        if not args.no_extras:
            # one large call: the whole 100k-sentence corpus four times over (400k sentences, ~45 MB in, ~300 MB of 24-byte records out)
            reps_c = 4
            utf8_1, offs_1 = pack_sentences(corpora[0])
            n_big = reps_c * len(corpora[0])
            utf8_all = np.tile(utf8_1, reps_c)
            offs_all = np.concatenate([[0]] + [offs_1[1:] + k * int(offs_1[-1]) for k in range(reps_c)]).astype(np.uint64)
            capall = int(offs_all[-1]) // 2 + n_big  # tokens <= chars + 1 per sentence; the text is 3 bytes per char
            for name, alloc in (("large_call_pageable", np.empty), ("large_call_pinned", pinned_empty)):
                u = alloc(utf8_all.shape, dtype=np.uint8); u[:] = utf8_all
                o = alloc(offs_all.shape, dtype=np.uint64); o[:] = offs_all
                big = (alloc(capall, dtype=TOKEN_DTYPE), alloc(n_big + 1, dtype=np.uint64), alloc(n_big, dtype=np.uint8))
                big[0].view(np.uint8)[::4096] = 0  # pages touched
                tok.tokenize_packed(u, o, out=big)  # untimed: scratch allocation, staging buffers, worker threads
                ts = []
                for _ in range(5):
                    t1 = time.perf_counter()
                    tok.tokenize_packed(u, o, out=big)
                    ts.append(time.perf_counter() - t1)
                ts.sort()
                result["pcie_inclusive"][name] = n_big / ts[len(ts) // 2]  # median of five calls
                result["pcie_inclusive"][name + "_calls_ms"] = [round(x * 1e3, 3) for x in ts]
                del big
            result["pcie_inclusive"]["large_call_what"] = (f"kgpu_tokenize_batch, ONE call over {n_big} sentences (the cfg 2 corpus x {reps_c}), host memory in, dense 24-byte "
                                                          "records out: chunks of <= 8192 sentences, 8-byte records written by the compaction kernel into mapped pinned "
                                                          "memory, expanded into the caller's buffer by worker threads while later chunks compute")
            result["value_end_to_end"] = {"value": max(result["pcie_inclusive"]["large_call_pageable"], result["pcie_inclusive"]["large_call_pinned"]),
                                          "unit": "sentences/s", "what": "SURVEY 8(d) end-to-end incl. H2D / D2H: the better of pcie_inclusive.large_call_{pageable,pinned}; "
                                                                         "`value` is the device-resident rate"}

    if world == 1:
        # ---- the reference's server shape: many host threads, ONE sentence per call (src/tokenizer.rs:16 is &self, Send + Sync; src/bin/kanpyo.rs:106-126).
        # Native threads (kgpu_debug_concurrent_callers: Python threads would measure the GIL); concurrent small calls share launches (the combiner).
        try:
            from kanpyo_amd.tokenizer import concurrent_callers

            utf8_c, offs_c = pack_sentences(corpora[0][:20000])
            cc = {}
            for nthr, calls in ((1, 400), (16, 300), (64, 300), (128, 200), (128, 2000)):
                concurrent_callers(tok, utf8_c, offs_c, nthr, 20)  # warm: contexts, pinned blocks
                tok.routing(reset=True)
                cs0 = cgroup_cpu_stat()
                r = concurrent_callers(tok, utf8_c, offs_c, nthr, calls)
                cs1 = cgroup_cpu_stat()
                rt = tok.routing()
                # the callers' own CPU time per call, and whether the cgroup's CPU quota throttled the process during the leg (a throttled period
                # stops every thread for the rest of its 100 ms: that, not the device, is what a p99 of tens of milliseconds means here)
                r["cpu_us_per_call"] = r.pop("caller_cpu_s") * 1e6 / max(r["calls"], 1)
                r["quota_throttled_periods"] = cs1.get("nr_throttled", 0) - cs0.get("nr_throttled", 0) if cs0 else None
                r["combined_calls"], r["combined_launches"], r["small_calls"] = rt["combined_calls"], rt["combined_launches"], rt["small_calls"]
                r["sentences_per_launch"] = r["sentences"] / max(rt["small_calls"] - rt["combined_calls"] + rt["combined_launches"], 1)
                cc[f"threads{nthr}" + ("_sustained" if calls >= 1000 else "")] = r   # (sustained: several of the quota's 100 ms periods long)
            cc["what"] = ("kgpu_tokenize_batch with n = 1 in a loop from N native host threads over the first 20k cfg 2 sentences; closed loop, so "
                          "sentences/s = threads / mean latency (Little): calls that arrive while another thread's small launch is being assembled "
                          "join it (leader / follower, <= 15 us window, <= 128 sentences)")
            cc["host_cpus"] = cpu_quota()
            result["pcie_inclusive"]["concurrent_callers"] = cc
        except Exception as e:
            print(f"concurrent_callers leg failed: {e}", file=sys.stderr)
    if world == 1:
        # ---- the host-side merge of the multi-device call, alone (no device): what kgpu_tokenize_batch_multi's calling thread + workers sustain on this box's CPUs
        try:
            from kanpyo_amd.tokenizer import merge_bench

            result["multi_merge"] = merge_bench(8, 8192, 32, reps=20)
            result["multi_merge"]["host_cpus"] = cpu_quota()
        except Exception as e:
            print(f"multi_merge leg failed: {e}", file=sys.stderr)
    if extras_dir:  # the corpus generator (a pure-Python loop on one core) starts only now: the host-side legs above share the box's CPU quota with nothing
        open(os.path.join(extras_dir, "go"), "w").close()

    # ---- CPU baseline (rank 0, N==1 only): the oracle restatement on the host cores, every sentence tokenized once
    if world == 1 and not args.no_cpu:
        from oracle import oracle

        orc = oracle.OracleTokenizer.from_dict(sd.dict)
        utf8, offs = pack_sentences(corpora[0])
        n_c = len(corpora[0])
        bufs = (np.zeros(int(offs[-1]) + n_c, dtype=oracle.TOKEN_DTYPE), np.zeros(n_c + 1, dtype=np.uint64))
        exp0 = orc.tokenize_batch(utf8, offs, 1, out=bufs, copy=False)  # untimed: pages touched
        done, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds:
            t1 = time.perf_counter()
            exp0 = orc.tokenize_batch(utf8, offs, 1, out=bufs, copy=False)
            t_cpu += time.perf_counter() - t1
            done += n_c
        ncores = os.cpu_count() or 1
        # bit-exact check of the GPU's batch 0 against the same sentences from the oracle
        n0 = int(exp0.offsets[BATCH])
        g_tok, g_off = sample_tokens
        exact = bool(np.array_equal(g_off.astype(np.uint64), exp0.offsets[: BATCH + 1])
                     and np.array_equal(g_tok.reshape(-1), exp0.tokens[:n0].view(np.int32).reshape(-1).astype(np.int32)))
        # all cores: per-sentence slots in one preallocated buffer (no allocator, no merge copy inside the timed call), threads claim
        # runs of 64 sentences, several passes per call so that starting the threads is paid once; hardware threads and physical cores both tried
        slots = (np.zeros(int(offs[-1]) + n_c, dtype=oracle.TOKEN_DTYPE), np.zeros(n_c, dtype=np.uint32))
        orc.tokenize_slots(utf8, offs, ncores, 1, out=slots)  # untimed: pages touched
        all_cores = None
        quota = cpu_quota()
        for nthr in sorted({min(ncores, quota), min(ncores, 2 * quota)}, reverse=True):  # as many threads as CPUs the cgroup grants, and twice that
            reps_all = max(4, int(2.0 * result_rate_guess(done / t_cpu, nthr) / n_c))
            t1 = time.perf_counter()
            orc.tokenize_slots(utf8, offs, nthr, reps_all, out=slots)
            t_all = (time.perf_counter() - t1) / reps_all
            cand = {"value": n_c / t_all, "cores": nthr, "passes": reps_all,
                    "what": "korc_tokenize_slots: one preallocated slot range per sentence, threads claim runs of 64 sentences"}
            if all_cores is None or cand["value"] > all_cores["value"]:
                all_cores = cand
        all_cores["scaling_vs_1thread"] = all_cores["value"] / (done / t_cpu)
        all_cores["cpu_quota_cores"] = quota
        all_cores["host_hardware_threads"] = ncores
        all_cores["note"] = (f"the container's cgroup grants {quota} CPUs of the host's {ncores} hardware threads (cpu.max): "
                             "the all-core figure is bounded by that quota, not by the oracle") if quota < ncores else "no CPU quota"
        result["cpu_baseline"] = {
            "value": done / t_cpu, "unit": "sentences/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"the same 100k-sentence cfg2 corpus, {done // n_c} pass(es), {t_cpu:.1f} s, single pass per sentence into a "
                      "preallocated worst-case buffer, oracle/kanpyo_oracle.c (CPU restatement of Kanpyo's algorithm, gcc -O2), single thread",
            "all_cores": all_cores,
            "gpu_batch0_bit_exact": exact,
        }
        result["speedup_vs_cpu_1thread"] = result["value"] / result["cpu_baseline"]["value"]

    # ---- the other single-GPU configs (SURVEY 8d cfg 3, cfg 5) as extra lines
    if world == 1 and not args.no_extras:
        eng.close()
        extra = []
        extras_orc = None
        if not args.no_cpu:  # the checker: first batch of each extra config against the oracle (not timed)
            from oracle import oracle as _orc

            extras_orc = _orc.OracleTokenizer.from_dict(sd.dict)
        cfg2_relax_per_s = result["value"] * result["work_per_sentence"]["E"]
        for kind, passes, lab in (("dense", 10, "cfg 2-shaped text (100k sentences, ~40 chars, batch 4096) over the DENSE variant of the 392k-record dictionary: natural lattice "
                                                "density (SURVEY 8a a15: N ~ 8-10 x C; more than eight predecessors at about half of the positions)"),
                                  ("cfg5", 40, "BASELINE configs[4] (cfg 5): 1k sentences of 2048 chars, each with a same-category run > 1024 chars"),
                                  ("cfg3", 2, f"BASELINE configs[2] (cfg 3): {args.cfg3_sentences} mixed-length (8-512 char) sentences incl. unknown-word path, batches of 65536"),
                                  ("cfg3@4096", 2, f"BASELINE configs[2] (cfg 3): the same {args.cfg3_sentences} sentences in batches of 4096 (one sentence per wavefront slot: "
                                                   "a launch lasts as long as its longest sentence)")):
            batch_x = BATCH
            if kind == "cfg3":
                batch_x = 65536
            elif kind == "cfg3@4096":
                kind = "cfg3"
            flag = os.path.join(extras_dir, kind + "_done.npy")
            t_wait = time.perf_counter()
            while not os.path.exists(flag) and extras_proc.is_alive() and time.perf_counter() - t_wait < 600:
                time.sleep(0.2)
            if not os.path.exists(flag):
                print(f"{kind}: corpus generator did not finish; skipped", file=sys.stderr)
                continue
            u = np.load(os.path.join(extras_dir, kind + "_utf8.npy"))
            o = np.load(os.path.join(extras_dir, kind + "_offs.npy"))
            n_chars = int(np.load(flag)[0])
            try:
                # cfg 3 names no batch size (BASELINE configs[2]): batches of 65536 -- a launch lasts as long as its longest sentence, and with one
                # sentence per wavefront slot (4096) a few 500-char sentences decide a launch whose average is 120 (round 3, M sentences/s by batch:
                # 4096 12.8, 16384 19.0-19.2, 32768 19.5, 65536 19.9)
                wl_x = PackedWorkload(u, o, batch=batch_x)
                if kind == "dense":
                    from kanpyo_amd.dict import Dict as _Dict

                    dd = _Dict.load_npz(os.path.join(extras_dir, "dense_dict.npz"))
                    tok_d = Tokenizer(dd, device=local_rank)
                    orc_d = _orc.OracleTokenizer.from_dict(dd) if extras_orc is not None else None
                    line = measure_config(tok_d, dev, wl_x, n_chars, passes, args.queue, 0, lab, orc=orc_d)
                    w = line["work_per_sentence"]
                    line["lattice_density"] = {"nodes_per_char": w["N"] / max(w["C"], 1e-9), "relaxations_per_node": w["E"] / max(w["N"], 1e-9),
                                               "cfg2_nodes_per_char": result["work_per_sentence"]["N"] / result["work_per_sentence"]["C"]}
                    line["relaxations_per_s"] = line["value"] * w["E"]
                    line["relaxations_per_s_vs_cfg2"] = line["relaxations_per_s"] / max(cfg2_relax_per_s, 1e-9)
                    line["routing_what"] = ("deferred[0]: sentences that left the pool kernel (more than 8 dictionary prefixes at one position -- MAXM -- or a "
                                            "lattice beyond the LDS routing limit); redone[0]: of which after the walk had been paid for")
                    tok_d.close()
                    extra.append(line)
                else:
                    extra.append(measure_config(tok, dev, wl_x, n_chars, passes, args.queue, 0, lab, orc=extras_orc))
                    if kind == "cfg5":   # the config as literally written: ONE batch of 1k documents at a time (the runtime gives such a list two wavefronts per document)
                        line = measure_config(tok, dev, wl_x, n_chars, 12, 1, 0, lab + " -- ONE batch in flight (one context)", orc=None)
                        line["contexts"] = 1
                        extra.append(line)
            except Exception as e:
                print(f"{kind} leg failed: {e}", file=sys.stderr)
        # ---- ONE context (a caller that keeps a single batch in flight): the cfg 2 corpus in batches of 4096 and of 16384
        try:
            u2, o2 = pack_sentences(corpora[0])
            chars2 = sum(map(len, corpora[0]))
            for b1 in (BATCH, 4 * BATCH):
                line = measure_config(tok, dev, PackedWorkload(u2, o2, batch=b1), chars2, 10, 1, 0,
                                      f"ONE context, one batch in flight: the cfg 2 corpus (100k sentences, ~40 chars) in batches of {b1}", orc=None)
                line["contexts"] = 1
                extra.append(line)
        except Exception as e:
            print(f"single-context leg failed: {e}", file=sys.stderr)
        # ---- a real Kanpyo dictionary, when one is on the box (KANPYO_DICT=/path/ipa.dict, optional KANPYO_SENTENCES=/path/text): parity + rate on it.
        # The dictionary cannot be obtained in the build environment (reference README.md:74-82: fetched from GitHub Releases), so this line is
        # normally absent; tests/test_real_dict.py is the matching parity test.
        real = os.environ.get("KANPYO_DICT")
        if real and os.path.exists(real):
            try:
                from kanpyo_amd.dictfile import load_dict

                df = load_dict(real)
                tok_r = Tokenizer(df.dict, device=local_rank)
                sp = os.environ.get("KANPYO_SENTENCES")
                if sp and os.path.exists(sp):
                    with open(sp, encoding="utf-8") as f:
                        rs = [ln.rstrip() for ln in f.read().split("\n") if ln.strip()]
                else:  # no text given: the synthetic cfg 2 corpus (its words are not this dictionary's: an unknown-word-heavy load)
                    rs = corpora[0]
                rs = (rs * (N_SENT // max(len(rs), 1) + 1))[:N_SENT]
                ur, orr = pack_sentences(rs)
                orc_r = _orc.OracleTokenizer.from_dict(df.dict) if extras_orc is not None else None
                line = measure_config(tok_r, dev, PackedWorkload(ur, orr, batch=BATCH), sum(map(len, rs)), 10, args.queue, 0,
                                      f"real dictionary {os.path.basename(real)} ({df.dict.n_morphs} records), {len(rs)} sentences, batch 4096", orc=orc_r)
                extra.append(line)
                tok_r.close()
            except Exception as e:
                print(f"real-dictionary leg failed: {e}", file=sys.stderr)
        result["extra"] = extra
        import shutil

        shutil.rmtree(extras_dir, ignore_errors=True)

    result["roofline"] = hoist_roofline(result["roofline"])
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(result), flush=True)
    if multi:
        os.dup2(2, 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

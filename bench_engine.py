#!/usr/bin/env python
"""bench_engine.py -- the job bench.py times: a rank's batches (Workload), the device contexts that keep Q of them in flight with
inputs and results resident in HBM (GpuEngine), and the loop over exactly K steps with the chunked gather of the multi-rank path
(run_job).  Importable without a GPU: tests/test_dist_cpu.py drives run_job / Workload with world size 2 over gloo and a CPU engine."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
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


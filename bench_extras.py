#!/usr/bin/env python
"""bench_extras.py -- what bench.py measures at N = 1 besides the headline: the per-stage split of the fused kernel (ablation levels), the instruction
roofline, the other single-GPU BASELINE configs (cfg 3, cfg 5, the dense-lattice dictionary, one context), the host-buffer entry point (PCIe-inclusive
rates, call latencies, concurrent one-sentence callers), the host-side merge of the multi-device call, and `bench.py --single-process` (cfg 4 through
kgpu_multi_* in one process).  Every leg returns a plain dict; bench.py writes them all to bench_full.json and puts a few scalars of each on its line."""
import json
import os
import sys
import time

import numpy as np

from bench_engine import (BATCH, CHIP_CLOCK_HZ, CHIP_SIMDS, HBM_PEAK_GBS, N_SENT, ROOT, GpuEngine, PackedWorkload, Workload, algorithmic_bytes,
                          cgroup_cpu_stat, cpu_model, cpu_quota, result_rate_guess, run_job)


def extras_child(sd, outdir, cfg3_n):
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
    return result



# ------------------------------------------------------------------ legs of the N = 1 run (each returns a dict for bench_full.json)

def measure_stages(eng, torch, stage_bytes, steps=3):
    """Per-stage roofline: the same pipeline with every sentence stopped after a stage (kgpu_ctx_set_ablation, measurement-only mode of the runtime);
    a stage's time is the difference of consecutive stop levels at full occupancy.  Five interleaved repetitions of `steps` steps per level, the FASTEST
    counts: a stopped chain is a 25-40 us kernel per 4096-sentence batch, so a level's time is easily the host's launch rate (15-20 us per batch) or one
    scheduling hiccup instead of the GPU's -- bench.py therefore hands over an engine with batches of 16 384 (a quarter of the launches per step).
    stage_bytes: (A, B, C) algorithmic bytes per step.  B_viterbi = connection-cost gather + sweep."""
    from kanpyo_amd.device import STAGE_ALL, STAGE_LATTICE, STAGE_VITERBI

    runs = {"A": [], "AB": [], "ABC": []}
    for _ in range(5):
        for name, stop in (("A", STAGE_LATTICE), ("AB", STAGE_VITERBI), ("ABC", STAGE_ALL)):
            for c in eng.ctxs:
                c.set_ablation(stop)
            run_job(eng, 1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_job(eng, steps)
            runs[name].append((time.perf_counter() - t1) / steps * 1e3)
    for c in eng.ctxs:
        c.set_ablation(STAGE_ALL)
    ms = {k: min(v) for k, v in runs.items()}
    sb = dict(zip(("A_lattice", "B_viterbi", "C_emit"), stage_bytes))
    sm = {"A_lattice": ms["A"], "B_viterbi": ms["AB"] - ms["A"], "C_emit": ms["ABC"] - ms["AB"]}

    def line(k):
        ok = sm[k] > 0 and sb[k] / (sm[k] * 1e-3) / 1e9 <= HBM_PEAK_GBS  # a difference of two timings can come out at or below zero: then it says nothing
        gbs = sb[k] / (sm[k] * 1e-3) / 1e9 if ok else None
        return {"bytes_per_step": sb[k], "ms_per_step": sm[k], "achieved": gbs, "frac": gbs / HBM_PEAK_GBS if ok else None}

    out = {k: line(k) for k in sb}
    out["level_ms_per_step_runs"] = {k: [round(x, 4) for x in v] for k, v in runs.items()}
    return out


def stage_split(eng, tok, dev, corpora, queue, torch, stage_bytes):
    """measure_stages over an engine of its own with batches of 16 384: in batches of 4096 the level that stops after the lattice is a 25 us kernel per batch
    against the submitting thread's 15-20 us per batch -- partly the host's rate, and stage B, a difference, came out at 0.21-0.36 ms by the box's CPU
    (profiles/experiments/r06_tile_sweep.txt).  If that engine cannot be made the headline's own (batches of 4096) is used: the split never breaks the line."""
    eng_s = None
    try:
        eng_s = GpuEngine(tok, dev, Workload(corpora[:1], 0, 1, batch=4 * BATCH), queue=queue, streams=0, ring=1)
        run_job(eng_s, 4)
        out = measure_stages(eng_s, torch, stage_bytes, steps=8)
        out["batch"] = 4 * BATCH
    except Exception as e:
        print(f"stage split in batches of {4 * BATCH} failed ({e}): batches of {BATCH}", file=sys.stderr)
        out = measure_stages(eng, torch, stage_bytes)
        out["batch"] = BATCH
    if eng_s is not None:
        eng_s.close()
    return out


def instruction_roofline(rate, kernel_source_hash):
    """The ceiling the pool kernel is actually near: VALU issue.  Instruction counts per sentence come from a separate rocprofv3 --pmc pass
    (profiles/pmc_instructions.json); the rate is this run's.  valu_issue_frac keeps its meaning of rounds 1-4 (2 cycles per wave op: the floor);
    valu_issue_frac_3cyc is the static-mix average (tools/ubench/valu.hip: DPP / VOP3 / v_min / v_cndmask forms take 4)."""
    ipath = os.path.join(ROOT, "profiles", "pmc_instructions.json")
    if not os.path.exists(ipath):
        return None
    ins = json.load(open(ipath))
    valu, salu = ins["valu_per_sentence"], ins["salu_per_sentence"]
    per = rate / (CHIP_SIMDS * CHIP_CLOCK_HZ)
    return {"valu_per_sentence": valu, "salu_per_sentence": salu, "lds_per_sentence": ins.get("lds_per_sentence"), "source": ins.get("source", "profiles/pmc_instructions.json"),
            "stale": ins.get("kernel_src_sha16") != kernel_source_hash(),
            "valu_issue_frac": valu * 2.0 * per, "valu_issue_frac_3cyc": valu * 3.0 * per, "valu_issue_frac_4cyc": valu * 4.0 * per}


def host_paths(tok, wl, cap, corpora, large=True):
    """kgpu_tokenize_batch (H2D + kernels + D2H per call): PCIe-inclusive rates and call latencies -- never `value`."""
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences, pinned_empty

    utf8_0, offs_0 = wl.packed[0][0]
    h_out = (np.empty(cap, dtype=TOKEN_DTYPE), np.empty(BATCH + 1, dtype=np.uint64), np.empty(BATCH, dtype=np.uint8))
    tok.tokenize_packed(utf8_0, offs_0, out=h_out)  # untimed: the pool ctx allocates its scratch, pages get touched
    per_call = []
    for i in range(min(wl.nb(0), 12) * 2):
        u, o = wl.packed[0][i % min(wl.nb(0), 12)]
        t1 = time.perf_counter()
        tok.tokenize_packed(u, o, out=h_out)
        per_call.append((time.perf_counter() - t1) / (len(o) - 1))
    per_call.sort()
    res = {"value": 1.0 / per_call[len(per_call) // 2], "unit": "sentences/s", "slowest_call_sentences_per_s": 1.0 / per_call[-1],
           "what": "kgpu_tokenize_batch, pageable host buffers, one 4096-sentence call at a time (median of 24 calls over 12 batches)"}
    lat = {}
    for n_call in (1, 64, 4096):  # the reference's call shape is n = 1: Tokenizer::tokenize(&str), once per CLI line (src/bin/kanpyo.rs:106-126)
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
        lat[f"n{n_call}"] = {"median_us": ts[len(ts) // 2] * 1e6, "p10_us": ts[len(ts) // 10] * 1e6, "sentences_per_s_at_median": n_call / ts[len(ts) // 2]}
    res["call_latency"] = lat
    if large:  # one large call: the whole 100k-sentence corpus four times over (400k sentences, ~45 MB in, ~300 MB of 24-byte records out)
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
            res[name] = n_big / ts[len(ts) // 2]  # median of five calls
            res[name + "_calls_ms"] = [round(x * 1e3, 3) for x in ts]
            del big
        res["large_call_sentences"] = n_big
    return res


def concurrent_callers_leg(tok, corpora):
    """The reference's server shape: many host threads, ONE sentence per call (src/tokenizer.rs:16 is &self, Send + Sync).  Native threads
    (kgpu_debug_concurrent_callers: Python threads would measure the GIL); concurrent small calls share launches (the combiner).  Closed loop."""
    from kanpyo_amd.tokenizer import concurrent_callers, pack_sentences

    utf8_c, offs_c = pack_sentences(corpora[0][:20000])
    cc = {}
    for nthr, calls in ((1, 400), (16, 300), (64, 300), (128, 200), (128, 2000)):
        concurrent_callers(tok, utf8_c, offs_c, nthr, 20)  # warm: contexts, pinned blocks
        tok.routing(reset=True)
        cs0 = cgroup_cpu_stat()
        r = concurrent_callers(tok, utf8_c, offs_c, nthr, calls)
        cs1 = cgroup_cpu_stat()
        rt = tok.routing()
        r["cpu_us_per_call"] = r.pop("caller_cpu_s") * 1e6 / max(r["calls"], 1)
        r["quota_throttled_periods"] = cs1.get("nr_throttled", 0) - cs0.get("nr_throttled", 0) if cs0 else None
        r["combined_calls"], r["combined_launches"], r["small_calls"] = rt["combined_calls"], rt["combined_launches"], rt["small_calls"]
        r["sentences_per_launch"] = r["sentences"] / max(rt["small_calls"] - rt["combined_calls"] + rt["combined_launches"], 1)
        cc[f"threads{nthr}" + ("_sustained" if calls >= 1000 else "")] = r  # (sustained: several of the quota's 100 ms periods long)
    cc["host_cpus"] = cpu_quota()
    return cc


def multi_merge_leg():
    """The host-side merge of the multi-device call, alone (no device): what kgpu_tokenize_batch_multi's calling thread + workers sustain on this box's CPUs,
    24-byte records out and -- kgpu_tokenize_batch_multi_compact -- 8-byte records out."""
    from kanpyo_amd.tokenizer import merge_bench

    out = merge_bench(8, 8192, 32, reps=20)
    try:
        out["compact"] = merge_bench(8, 8192, 32, reps=20, compact=True)
    except TypeError:
        pass
    out["host_cpus"] = cpu_quota()
    return out


EXTRA_CONFIGS = (  # key, corpus kind, passes, batch, contexts, workload label
    ("dense", "dense", 10, BATCH, None, "cfg 2-shaped text (100k sentences, ~40 chars, batch 4096) over the DENSE variant of the 392k-record dictionary: natural lattice density (SURVEY 8a a15: N ~ 8-10 x C)"),
    ("cfg5_q8", "cfg5", 40, BATCH, None, "BASELINE configs[4] (cfg 5): 1k sentences of 2048 chars, each with a same-category run > 1024 chars; eight batches in flight"),
    ("cfg5_one_batch", "cfg5", 12, BATCH, 1, "BASELINE configs[4] (cfg 5) as written: ONE batch of 1k documents in flight (one context)"),
    ("cfg3_b65536", "cfg3", 2, 65536, None, "BASELINE configs[2] (cfg 3): mixed-length (8-512 char) sentences incl. unknown-word path, batches of 65536"),
    ("cfg3_b4096", "cfg3", 2, BATCH, None, "BASELINE configs[2] (cfg 3): the same sentences in batches of 4096"),
)


def other_configs(tok, sd, dev, local_rank, args, extras_dir, extras_proc, corpora, cfg2_line, with_oracle):
    """cfg 3, cfg 5, the dense dictionary, one context: {key: measure_config line}.  The first batch of each is compared with the oracle (untimed)."""
    from kanpyo_amd import Tokenizer
    from kanpyo_amd.tokenizer import pack_sentences

    orc_mod = None
    if with_oracle:  # the checker (not timed)
        from oracle import oracle as orc_mod
    orc = orc_mod.OracleTokenizer.from_dict(sd.dict) if orc_mod else None
    out, loaded, tok_d, orc_d = {}, {}, None, None
    for key, kind, passes, batch_x, ctxs, lab in EXTRA_CONFIGS:
        flag = os.path.join(extras_dir, kind + "_done.npy")
        t_wait = time.perf_counter()
        while not os.path.exists(flag) and extras_proc.is_alive() and time.perf_counter() - t_wait < 600:
            time.sleep(0.2)
        if not os.path.exists(flag):
            print(f"{kind}: corpus generator did not finish; skipped", file=sys.stderr)
            continue
        if kind not in loaded:
            loaded[kind] = (np.load(os.path.join(extras_dir, kind + "_utf8.npy")), np.load(os.path.join(extras_dir, kind + "_offs.npy")), int(np.load(flag)[0]))
        u, o, n_chars = loaded[kind]
        try:
            wl_x = PackedWorkload(u, o, batch=batch_x)
            if kind == "dense":
                from kanpyo_amd.dict import Dict as _Dict

                dd = _Dict.load_npz(os.path.join(extras_dir, "dense_dict.npz"))
                tok_d = Tokenizer(dd, device=local_rank)
                orc_d = orc_mod.OracleTokenizer.from_dict(dd) if orc_mod else None
                line = measure_config(tok_d, dev, wl_x, n_chars, passes, args.queue, 0, lab, orc=orc_d)
                w = line["work_per_sentence"]
                line["lattice_density"] = {"nodes_per_char": w["N"] / max(w["C"], 1e-9), "relaxations_per_node": w["E"] / max(w["N"], 1e-9),
                                           "cfg2_nodes_per_char": cfg2_line["work_per_sentence"]["N"] / cfg2_line["work_per_sentence"]["C"]}
                line["relaxations_per_s"] = line["value"] * w["E"]
                line["relaxations_per_s_vs_cfg2"] = line["relaxations_per_s"] / max(cfg2_line["value"] * cfg2_line["work_per_sentence"]["E"], 1e-9)
                # the dense leg's own stage split (its E is 2.6x cfg 2's: gather + sweep dominate harder) and its own single-thread CPU rate
                import torch

                eng_d = GpuEngine(tok_d, dev, wl_x, queue=args.queue, streams=0, ring=1)
                run_job(eng_d, 3)
                a_, b_, c_ = algorithmic_bytes({k: line["work_per_sentence"][k] * line["sentences"] for k in ("B", "C", "T", "N", "E", "K")})
                line["stages"] = measure_stages(eng_d, torch, (a_, b_, c_))
                eng_d.close()
                if orc_d is not None:
                    line["cpu_1thread"] = cpu_rate_pinned(orc_d, u, o, passes=3)
                tok_d.close()
            else:
                line = measure_config(tok, dev, wl_x, n_chars, passes, ctxs or args.queue, 0, lab, orc=orc if ctxs is None else None)
                if ctxs:
                    line["contexts"] = ctxs
                if key == "cfg3_b65536" and orc is not None:  # cfg 3's own single-thread CPU rate, on its first 50k sentences
                    n_s = min(50_000, len(o) - 1)
                    line["cpu_1thread"] = cpu_rate_pinned(orc, u[: int(o[n_s])], o[: n_s + 1], passes=3)
            out[key] = line
        except Exception as e:
            print(f"{key} leg failed: {e}", file=sys.stderr)
    # ---- ONE context (a caller that keeps a single batch in flight): the cfg 2 corpus in batches of 4096 and of 16384
    try:
        u2, o2 = pack_sentences(corpora[0])
        chars2 = sum(map(len, corpora[0]))
        for b1 in (BATCH, 4 * BATCH):
            line = measure_config(tok, dev, PackedWorkload(u2, o2, batch=b1), chars2, 10, 1, 0, f"ONE context, one batch in flight: the cfg 2 corpus in batches of {b1}", orc=None)
            line["contexts"] = 1
            out[f"one_ctx_b{b1}"] = line
    except Exception as e:
        print(f"single-context leg failed: {e}", file=sys.stderr)
    # ---- a real Kanpyo dictionary, when one is on the box (KANPYO_DICT=/path/ipa.dict, optional KANPYO_SENTENCES=/path/text): parity + rate on it.
    # The dictionary cannot be obtained in the build environment (reference README.md:74-82), so this line is normally absent (tests/test_real_dict.py).
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
            orc_r = orc_mod.OracleTokenizer.from_dict(df.dict) if orc_mod else None
            out["real_dict"] = measure_config(tok_r, dev, PackedWorkload(ur, orr, batch=BATCH), sum(map(len, rs)), 10, args.queue, 0,
                                              f"real dictionary {os.path.basename(real)} ({df.dict.n_morphs} records), {len(rs)} sentences, batch 4096", orc=orc_r)
            tok_r.close()
        except Exception as e:
            print(f"real-dictionary leg failed: {e}", file=sys.stderr)
    return out


# ------------------------------------------------------------------ the CPU baseline (the oracle restatement as the thing timed -- its allowed role)

def pin_to_one_core():
    """Pins the calling thread to ONE CPU of its affinity mask (the highest: away from CPU 0's interrupts); returns the old mask (None: not pinned)."""
    try:
        old = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {max(old)})
        return old
    except (AttributeError, OSError):
        return None


def cpu_rate_pinned(orc, utf8, offs, passes=5, out=None):
    """Single-thread rate of the oracle over (utf8, offs): thread pinned to one core, `passes` passes, the BEST one counts
    (a pass that shared its core with something else is slower, never faster); spread = (slowest - fastest) / fastest."""
    from oracle import oracle

    n = len(offs) - 1
    bufs = out or (np.zeros(int(offs[-1]) + n, dtype=oracle.TOKEN_DTYPE), np.zeros(n + 1, dtype=np.uint64))
    old = pin_to_one_core()
    try:
        orc.tokenize_batch(utf8, offs, 1, out=bufs, copy=False)  # untimed: pages touched
        ts = []
        for _ in range(passes):
            t1 = time.perf_counter()
            orc.tokenize_batch(utf8, offs, 1, out=bufs, copy=False)
            ts.append(time.perf_counter() - t1)
    finally:
        if old is not None:
            os.sched_setaffinity(0, old)
    return {"value": n / min(ts), "unit": "sentences/s", "passes": passes, "passes_spread": (max(ts) - min(ts)) / min(ts), "pinned": old is not None,
            "seconds": sum(ts), "sentences": n}


def cpu_baseline_leg(sd, corpora, sample_tokens, passes):
    """The oracle restatement (oracle/kanpyo_oracle.c, gcc -O2) on the host cores: one thread pinned to one core, best of `passes` passes over the
    100k-sentence corpus (~1.2 s each); then all the cores the cgroup grants.  Also the bit-exact check of the GPU's batch 0."""
    from kanpyo_amd.tokenizer import pack_sentences
    from oracle import oracle

    orc = oracle.OracleTokenizer.from_dict(sd.dict)
    utf8, offs = pack_sentences(corpora[0])
    n_c = len(corpora[0])
    bufs = (np.zeros(int(offs[-1]) + n_c, dtype=oracle.TOKEN_DTYPE), np.zeros(n_c + 1, dtype=np.uint64))
    one = cpu_rate_pinned(orc, utf8, offs, passes=passes, out=bufs)
    exp0 = orc.tokenize_batch(utf8[: int(offs[BATCH])], offs[: BATCH + 1], 1)  # batch 0 again, for the check (untimed)
    g_tok, g_off = sample_tokens
    n0 = int(exp0.offsets[BATCH])
    exact = bool(np.array_equal(g_off.astype(np.uint64)[: BATCH + 1], exp0.offsets[: BATCH + 1])
                 and np.array_equal(g_tok.reshape(-1), exp0.tokens[:n0].view(np.int32).reshape(-1)))
    # all cores: per-sentence slots in one preallocated buffer, threads claim runs of 64 sentences, several passes per call
    ncores, quota = os.cpu_count() or 1, cpu_quota()
    slots = (np.zeros(int(offs[-1]) + n_c, dtype=oracle.TOKEN_DTYPE), np.zeros(n_c, dtype=np.uint32))
    orc.tokenize_slots(utf8, offs, ncores, 1, out=slots)  # untimed: pages touched
    all_cores = None
    for nthr in sorted({min(ncores, quota), min(ncores, 2 * quota)}, reverse=True):
        reps_all = max(4, int(2.0 * result_rate_guess(one["value"], nthr) / n_c))
        t1 = time.perf_counter()
        orc.tokenize_slots(utf8, offs, nthr, reps_all, out=slots)
        cand = {"value": n_c / ((time.perf_counter() - t1) / reps_all), "cores": nthr, "passes": reps_all}
        if all_cores is None or cand["value"] > all_cores["value"]:
            all_cores = cand
    all_cores.update(scaling_vs_1thread=all_cores["value"] / one["value"], cpu_quota_cores=quota, host_hardware_threads=ncores)
    return {"value": one["value"], "unit": "sentences/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(), "passes": one["passes"],
            "passes_spread": one["passes_spread"], "pinned": one["pinned"],
            "sample": f"cfg 2 corpus (100k sentences), best of {one['passes']} passes ({one['seconds']:.1f} s), 1 thread pinned, oracle/kanpyo_oracle.c gcc -O2",
            "all_cores": all_cores, "gpu_batch0_bit_exact": exact}

#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: sentences/sec (+ input MiB/s) of Tokenizer::tokenize on MI355X, synthetic IPADIC-shaped dictionary.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A STEP is one pass of the hot path over one whole 100 000-sentence corpus in batches of 4096 (24 full batches and the 1 696-sentence tail),
inputs resident in HBM, dense token streams left in HBM.

  N = 1   BASELINE configs[1] (SURVEY 8d cfg 2): the seed-1 corpus, every step.
  N > 1   BASELINE configs[3] (cfg 4): step k is the corpus of seed 100 + k (cycled), sentence i -> GPU i mod N, dictionary replicated, no data-path
          collective; the token records of every step are gathered to rank 0 (flat gatherv over xGMI) inside the timed region, so `value` is the whole
          job's rate and `scaling` is "strong" (the work of a step does not grow with N).

Rank 0 prints ONE compact JSON line (<= 6 KB: compact_line) as the LAST line of stdout and writes everything it measured to bench_full.json (next to this
script, and under gpurun_out/ when that directory exists).  At N = 1 the line carries the per-launch roofline of the dominant kernel (HIP events on the
kernel's stream), the per-stage fractions (runtime ablation mode), the CPU baseline (the oracle restatement pinned to one core, best of five passes) and a
few scalars of every extra leg (bench_extras.py: cfg 3, cfg 5, the dense-lattice dictionary, one context, host-buffer calls).

The job itself (Workload / GpuEngine / run_job) lives in bench_engine.py and is re-exported here: tests/test_dist_cpu.py drives it over gloo.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from bench_engine import (BATCH, HBM_PEAK_GBS, N_SENT, ROOT, GpuEngine, PackedWorkload, Workload, algorithmic_bytes, c_getenv,  # noqa: F401 (re-exported)
                          chunk_steps_for, cpu_model, cpu_quota, expand_gathered, result_rate_guess, run_job)

LINE_LIMIT = 6144  # bytes: the driver's record of round 5 lost a 24 KB line (BENCH_r05.json parsed: null)
KERNEL = "k_tokenize_pool (fused lattice build + Viterbi + backtrace, LDS page pool)"


def _r(v, digits=5):
    """Numbers on the line with `digits` significant digits (floats only; ints, bools, None and strings pass)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float(f"{v:.{digits}g}")


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d and not isinstance(d[k], (dict, list))}


def compact_line(full):
    """The line the driver parses: scalars only below the second level, no prose.  `full` is the complete report (bench_full.json)."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                        "input_MiB_per_s", "sentences_total", "speedup_vs_cpu_1thread", "speedup_vs_1gpu"))
    cfg = dict(full.get("config") or {})
    cfg["workload"] = str(cfg.get("workload", ""))[:118]
    line["config"] = {k: (v if isinstance(v, str) else _r(v)) for k, v in cfg.items() if not isinstance(v, (dict, list)) or k == "devices"}
    r = full.get("roofline") or {}
    line["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "stage_A_frac", "stage_B_frac", "stage_C_frac", "stage_B_ms",
                                 "frac_at_job_rate", "frac_alone", "avg_kernel_ms", "kernel_alone_ms", "kernel", "valu_issue_frac", "valu_issue_frac_3cyc",
                                 "insts_per_sentence", "peak_measured_read", "frac_of_measured_read", "algorithmic_bytes_per_launch",
                                 "algorithmic_bytes_per_sentence", "traffic_stale", "launches_in_flight"))
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "passes", "passes_spread", "pinned", "gpu_batch0_bit_exact"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:118]
        if cb.get("all_cores"):
            line["cpu_baseline"]["all_cores_value"], line["cpu_baseline"]["all_cores"] = _r(cb["all_cores"]["value"]), cb["all_cores"]["cores"]
    if isinstance(full.get("value_end_to_end"), dict):
        line["value_end_to_end"] = _r(full["value_end_to_end"]["value"])
    ex = full.get("extra") or {}
    if "dense" in ex:
        line["value_dense"] = _r(ex["dense"]["value"])
        line["stage_B_frac_dense"] = _r(((ex["dense"].get("stages") or {}).get("B_viterbi") or {}).get("frac"))
    summ = {}
    for k, e in ex.items():
        s = {"value": _r(e["value"]), "frac_at_job_rate": _r(e["roofline_at_job_rate"]["frac"]), "slot_occupancy": _r(e.get("slot_occupancy"), 3),
             "bit_exact": e.get("first_batch_bit_exact_vs_oracle")}
        if k.startswith("cfg"):
            s["Gchar_per_s"] = _r(e["Mchar_per_s"] / 1e3, 4)
        if "cpu_1thread" in e:
            s["cpu_1thread"] = _r(e["cpu_1thread"]["value"], 4)
        summ[k] = s
    pc = full.get("pcie_inclusive") or {}
    if "call_latency" in pc:
        summ["call_latency_us"] = {k: _r(v["median_us"], 4) for k, v in pc["call_latency"].items()}
    if "concurrent_callers" in pc:
        summ["callers_sentences_per_s"] = {k: _r(v["sentences_per_s"], 4) for k, v in pc["concurrent_callers"].items() if isinstance(v, dict) and "sentences_per_s" in v}
    mm = full.get("multi_merge")
    if mm:
        summ["multi_merge"] = {"sentences_per_s": _r(mm.get("sentences_per_s"), 4), "compact_sentences_per_s": _r((mm.get("compact") or {}).get("sentences_per_s"), 4),
                               "host_cpus": mm.get("host_cpus")}
    if summ:
        line["extra_summary"] = summ
    g = full.get("gather")
    if g:
        line["gather"] = _pick(g, ("chunks", "tokens", "sentences", "complete", "reassembled_step_equals_one_gpu", "chunk_steps", "record_bytes", "root_ingest_GB_per_s"))
    if full.get("per_rank"):
        line["per_rank_sentences_per_s"] = [_r(p.get("sentences_per_s"), 4) for p in full["per_rank"]]
    if full.get("one_gpu_leg"):
        line["one_gpu_value"] = _r(full["one_gpu_leg"]["value"])
    line["full_report"] = "bench_full.json"
    return line


def fit_line(line, limit=LINE_LIMIT):
    """-> the line as compact JSON text of at most `limit` bytes: what goes first if it ever grows too long is the least needed (never the contract's keys)."""
    text = json.dumps(line, separators=(",", ":"))
    for victim in ("per_rank_sentences_per_s", "extra_summary", "gather", "config"):
        if len(text) <= limit:
            break
        if victim == "config":
            line["config"] = {"workload": line["config"].get("workload", "")}
        else:
            line.pop(victim, None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(full, real_stdout=None):
    """bench_full.json (beside the script, and under gpurun_out/ where that exists), then the compact line as the LAST line of stdout."""
    text = json.dumps(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(text + "\n")
            except OSError as e:
                print(f"bench_full.json not written to {d}: {e}", file=sys.stderr)
    line = fit_line(compact_line(full))
    sys.stdout.flush()
    if real_stdout is not None:
        os.dup2(real_stdout, 1)
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queue", type=int, default=8, help="batches in flight (one context each)")
    ap.add_argument("--streams", type=int, default=4, help="N>1: HIP streams the contexts share round-robin")
    ap.add_argument("--corpora", type=int, default=0, help="N>1: distinct cfg 4 corpora (seeds 100..) generated and cycled; 0 = one per step, at most 100")
    ap.add_argument("--no-one-gpu-leg", action="store_true", help="N>1: skip the untimed leg in which rank 0 runs the same steps alone (speedup_vs_1gpu)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5, help="untimed: the same steps for this long before the W warmup steps")
    ap.add_argument("--cpu-passes", type=int, default=5, help="CPU baseline: passes over the corpus, the best counts")
    ap.add_argument("--cfg3-sentences", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg 3 / cfg 5 / dense / latency legs (the stage split stays)")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-stage split too")
    ap.add_argument("--gather-records", type=int, default=8, choices=(8, 24), help="N>1: bytes per token record on the wire (8 = kgpu_token8)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (self-test)")
    ap.add_argument("--single-process", action="store_true", help="--gpus N in ONE process through kgpu_multi_* (no torch.distributed)")
    ap.add_argument("--devices", default="", help="--single-process: the device of every entry, e.g. 0,0 (default: entry g -> device g)")
    args = ap.parse_args()

    # stdout carries the result line only: everything else that libraries print there (RCCL's banner ...) goes to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.single_process:
        import bench_extras

        emit(bench_extras.run_single_process(args), real_stdout)
        return

    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    multi = world > 1 or args.force_dist
    K, W = args.steps, args.warmup

    import torch  # before libkanpyo_gpu.so (build_dict loads it): torch bundles its own libamdhip64 and must win
    import torch.distributed as dist

    from kanpyo_amd import _lib, synth

    _lib.lib()  # before any HIP call of this process: the library asks for the hardware queues it needs (kgpu_api.cpp: kgpu_preinit)
    sd = synth.build_dict()
    extras_dir = extras_proc = None
    if world == 1 and not args.no_extras:  # before the HIP runtime is initialised in this process: fork is safe
        import multiprocessing as mp
        import tempfile

        import bench_extras

        extras_dir = tempfile.mkdtemp(prefix="kanpyo_bench_")
        extras_proc = mp.get_context("fork").Process(target=bench_extras.extras_child, args=(sd, extras_dir, args.cfg3_sentences), daemon=True)
        extras_proc.start()

    assert torch.cuda.is_available(), "bench.py needs an MI355X: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=dev)

    from kanpyo_amd import Tokenizer
    from kanpyo_amd._lib import kernel_source_hash
    from kanpyo_amd.device import PROFILE_EVENTS, PROFILE_OFF, PROFILE_SAMPLED, PROFILE_WORK
    from kanpyo_amd.dist import ChunkedGather, reassemble

    # ---- workload
    if world == 1:
        corpora = [synth.make_corpus(sd, N_SENT, seed=1, kind="cfg2")]
        label = "BASELINE configs[1] (cfg 2): 100k synthetic ~40-char sentences, batch 4096, IPADIC-shaped 392k-record dictionary"
    else:
        ncorp = max(1, min(args.corpora if args.corpora > 0 else 100, 100, max(K, 1)))  # distinct seeds for every timed step (SURVEY 8d cfg 4)
        corpora = [synth.make_corpus(sd, N_SENT, seed=100 + k, kind="cfg2") for k in range(ncorp)]
        label = f"BASELINE configs[3] (cfg 4): 100k-sentence corpora seeds 100..{99 + ncorp}, sentence i -> GPU i mod {world}, gathered to rank 0"
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
            full1 = GpuEngine(tok, dev, Workload(corpora[:1], 0, 1), queue=2, streams=0, ring=1)
            for b in range(full1.nb(0)):
                full1.enqueue(0, b)
            fv, fc = full1.results(0)
            f_tok = torch.cat(fv).cpu().numpy()
            f_off = np.concatenate([[0], np.cumsum(fc.cpu().numpy())])
            gather_check = bool(np.array_equal(g_off, f_off) and np.array_equal(g_tok, f_tok))
            full1.close()
            del full1
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
    # HIP events around every 4th launch chain of every context, on the stream the kernels run on (what profiles/r06_kernel_stats.csv is compared with).  They are
    # not free -- a record is a barrier packet on its stream: five interleaved runs 124.5-136.6 M sentences/s against 136.9-138.5 with none and 136.7-137.8 with the
    # events on two of the eight contexts (BENCH_EVENT_CTXS=0,5: the per-launch fraction FALLS there while the rate rises; profiles/experiments/r06_tile_sweep.txt).
    ev = os.environ.get("BENCH_EVENT_CTXS", "all")   # (measurement: "all", a list like "0,5", or "" for none)
    timed_ctxs = set(range(len(eng.ctxs))) if ev == "all" else {int(x) % len(eng.ctxs) for x in ev.split(",") if x.strip()}
    for i, c in enumerate(eng.ctxs):
        if i in timed_ctxs and not os.environ.get("BENCH_NO_EVENTS"):
            c.set_profiling(PROFILE_EVENTS | PROFILE_SAMPLED)
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
    per_sentence_bytes = (a + b + c_) / n_work
    per_launch_bytes = per_sentence_bytes * BATCH
    avg_kernel_s = prof["first_ms"] / max(prof["launches"], 1) / 1e3    # the dominant kernel's own launches (what rocprofv3 reports for it)
    # every 4th launch chain is timed, tail batches included: scale the bytes to the average timed launch
    avg_launch_sentences = wl.sentences(0) / wl.nb(0)
    achieved = per_sentence_bytes * avg_launch_sentences / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    job_rate_bytes = per_sentence_bytes * sentences / elapsed / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from a separate rocprofv3 --pmc pass (tools/make_traffic_json.py)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = None
    plan = eng.ctxs[0].plan()
    full = {
        "metric": "sentences/sec", "value": sentences / elapsed, "unit": "sentences/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {"workload": label, "batch": BATCH, "sentences_per_step": N_SENT, "batches_per_step_per_gpu": wl.nb(0), "batches_in_flight": Q,
                   "streams": plan["streams"], "long_streams": plan["long_streams"], "GPU_MAX_HW_QUEUES": c_getenv("GPU_MAX_HW_QUEUES"),
                   "residency": "inputs resident in HBM, dense tokens left in HBM",
                   "sharding": f"sentence i -> GPU i mod N, dictionary replicated, one gatherv to rank 0 per {cs} step(s)" if multi else "single GPU"},
        "input_MiB_per_s": bytes_in / elapsed / 2**20,
        "work_per_sentence": {k: work[k] / n_work for k in ("B", "C", "T", "N", "E", "K")},
        "routing": {k: prof[k] for k in ("batches", "sentences", "deferred", "redone", "long_launches", "arena_regrows")},
        "launch_plan": plan,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
            "traffic_stale": (traffic.get("kernel_src_sha16") != kernel_source_hash()) if traffic else None,  # counters measured on other kernel sources than these
            "traffic_source": traffic.get("source") if traffic else None,
            "kernel": KERNEL, "algorithmic_bytes_per_sentence": per_sentence_bytes, "algorithmic_bytes_per_launch": per_launch_bytes,
            "stage_bytes_per_launch": {"A_lattice": a / n_work * BATCH, "B_viterbi": b / n_work * BATCH, "C_emit": c_ / n_work * BATCH},
            "avg_kernel_ms": avg_kernel_s * 1e3, "launches_timed": prof["launches"],
            "avg_launch_chain_ms": prof["tokenize_ms"] / max(prof["launches"], 1), "aux_kernels_avg_ms": prof["aux_ms"] / max(prof["launches"], 1),
            "launches_in_flight": Q, "achieved_at_job_rate": job_rate_bytes, "frac_at_job_rate": job_rate_bytes / HBM_PEAK_GBS,
        },
        "sentences_total": sentences,
    }
    roof = full["roofline"]
    if multi:
        full["gather"] = {"chunks": gathered["chunks"], "tokens": gathered["tokens"], "sentences": gathered["sentences"],
                          "complete": gathered["sentences"] == sentences, "reassembled_step_equals_one_gpu": gather_check,
                          "chunk_steps": cs, "record_bytes": 8 if compact else 24,
                          "root_ingest_GB_per_s": gathered["tokens"] * (8 if compact else 24) * (world - 1) / max(world, 1) / elapsed / 1e9}
        full["per_rank"] = per_rank
        full["corpora"] = {"distinct": len(corpora), "seeds": f"100..{99 + len(corpora)}", "sentences_each": N_SENT}
        if one_gpu is not None:
            full["one_gpu_leg"] = one_gpu
            full["speedup_vs_1gpu"] = full["value"] / one_gpu["value"]
        assert gathered["sentences"] == sentences, (gathered, sentences)
        assert gather_check, "gathered + reassembled token stream differs from the single-GPU stream"

    if world == 1:
        # ---- the dominant kernel alone on the chip: one full batch at a time, HIP events around every launch
        full_batches = [i for i in range(wl.nb(0)) if len(wl.packed[0][i][1]) - 1 == BATCH]
        c0 = eng.ctxs[0]
        c0.set_profiling(PROFILE_EVENTS)
        for _ in range(2):
            c0.profile(reset=True)
            for bi in full_batches:
                d_utf8, d_off, n, total = eng.inputs[0][bi]
                t, o, st = eng.out[0][bi]
                c0.tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, total, t.data_ptr(), eng.cap, o.data_ptr(), st.data_ptr())
                c0.sync()
        p = c0.profile(reset=True)
        c0.set_profiling(PROFILE_OFF)
        alone_ms = p["first_ms"] / max(p["launches"], 1)
        roof["kernel_alone_ms"] = alone_ms
        roof["frac_alone"] = per_launch_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if alone_ms > 0 else None
        # ---- the box's own streaming-read bandwidth (a 4 GiB int64 reduction, best of 6): second denominator of the roofline
        try:
            x = torch.empty(1 << 29, dtype=torch.int64, device=dev).fill_(1)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 0.0
            for _ in range(6):
                ev0.record(); x.sum(); ev1.record(); ev1.synchronize()
                best = max(best, x.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9)
            del x
            roof["peak_measured_read"], roof["frac_of_measured_read"] = best, achieved / best
        except Exception as e:  # never let the auxiliary measurement break the bench line
            print(f"streaming-read measurement skipped: {e}", file=sys.stderr)

    if world == 1 and not args.no_stages:
        import bench_extras

        roof["stages"] = bench_extras.stage_split(eng, tok, dev, corpora, args.queue, torch, (a, b, c_))
        for k, name in (("A", "A_lattice"), ("B", "B_viterbi"), ("C", "C_emit")):
            roof[f"stage_{k}_ms"], roof[f"stage_{k}_frac"] = roof["stages"][name]["ms_per_step"], roof["stages"][name]["frac"]
        try:
            ins = bench_extras.instruction_roofline(full["value"], kernel_source_hash)
            if ins:
                roof["instruction"] = ins
                roof["valu_issue_frac"], roof["valu_issue_frac_3cyc"] = ins["valu_issue_frac"], ins["valu_issue_frac_3cyc"]
                roof["insts_per_sentence"] = ins["valu_per_sentence"] + ins["salu_per_sentence"] + (ins.get("lds_per_sentence") or 0)
        except Exception as e:
            print(f"instruction roofline skipped: {e}", file=sys.stderr)

    if world == 1 and not args.no_extras:
        import bench_extras

        for name, leg in (("pcie_inclusive", lambda: bench_extras.host_paths(tok, wl, eng.cap, corpora)), ("multi_merge", bench_extras.multi_merge_leg)):
            try:
                full[name] = leg()
            except Exception as e:
                print(f"{name} leg failed: {e}", file=sys.stderr)
        try:
            full.setdefault("pcie_inclusive", {})["concurrent_callers"] = bench_extras.concurrent_callers_leg(tok, corpora)
        except Exception as e:
            print(f"concurrent_callers leg failed: {e}", file=sys.stderr)
        pc = full.get("pcie_inclusive") or {}
        if "large_call_pageable" in pc:  # SURVEY 8(d) end-to-end incl. H2D / D2H: the better of the two large host calls; `value` is the device-resident rate
            full["value_end_to_end"] = {"value": max(pc["large_call_pageable"], pc["large_call_pinned"]), "unit": "sentences/s"}
    if extras_dir:  # the corpus generator (a pure-Python loop on one core) starts only now: the host-side legs above share the box's CPU quota with nothing
        open(os.path.join(extras_dir, "go"), "w").close()

    # ---- CPU baseline (rank 0, N == 1 only)
    if world == 1 and not args.no_cpu:
        import bench_extras

        full["cpu_baseline"] = bench_extras.cpu_baseline_leg(sd, corpora, sample_tokens, args.cpu_passes)
        full["speedup_vs_cpu_1thread"] = full["value"] / full["cpu_baseline"]["value"]

    # ---- the other single-GPU configs (SURVEY 8d cfg 3, cfg 5), the dense dictionary, one context
    if world == 1 and not args.no_extras:
        import shutil

        eng.close()
        full["extra"] = bench_extras.other_configs(tok, sd, dev, local_rank, args, extras_dir, extras_proc, corpora, full, with_oracle=not args.no_cpu)
        shutil.rmtree(extras_dir, ignore_errors=True)

    emit(full, real_stdout)
    if multi:
        os.dup2(2, 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

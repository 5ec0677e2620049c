#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: sentences/sec (+ input MiB/s) of
Tokenizer::tokenize on MI355X, synthetic IPADIC-shaped dictionary.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of 4096 sentences
(BASELINE.json configs[1]: 100k synthetic ~40-char sentences, batch=4096), with
the batch already resident in HBM and the dense token stream left in HBM.
Multi-GPU is weak scaling: every rank owns its own 100k-sentence shard
(sentences shard with no data-path collective); the only communication is one
gatherv of the token records to rank 0 at the end of the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
N_SENT = 100_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(w):
    """Stage A (lattice build) B+16T+C+16N, Stage B (Viterbi) 8E+14N,
    Stage C (backtrace+emit) 28K -- all three run inside the one fused kernel."""
    a = w["B"] + 16 * w["T"] + w["C"] + 16 * w["N"]
    b = 8 * w["E"] + 14 * w["N"]
    c = 28 * w["K"]
    return a, b, c


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--queue", type=int, default=6, help="batches in flight (one context each)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the contexts share round-robin (HIP multiplexes streams onto "
                    "3 hardware queues: a 4th stream queues behind the 1st and unbalances them)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="lower bound of CPU-baseline work")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (self-test)")
    args = ap.parse_args()

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

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs an MI355X: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from kanpyo_amd import Tokenizer, synth
    from kanpyo_amd.device import PROFILE_EVENTS, PROFILE_OFF, PROFILE_SAMPLED, PROFILE_WORK, DeviceContext
    from kanpyo_amd.dist import ChunkedGather
    from kanpyo_amd.tokenizer import pack_sentences

    # ---- workload: dictionary replicated per GPU, one 100k-sentence shard per rank
    sd = synth.build_dict()
    corpus = synth.make_corpus(sd, N_SENT, seed=1 if world == 1 else 100 + rank, kind="cfg2")
    tok = Tokenizer(sd.dict, device=local_rank)
    batches = []
    for lo in range(0, N_SENT - BATCH + 1, BATCH):  # the 24 full batches (the 1696-sentence tail is a parity-test case)
        utf8, offs = pack_sentences(corpus[lo : lo + BATCH])
        batches.append((torch.from_numpy(utf8.copy()).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev),
                        int(offs[-1]), utf8, offs))
    nb = len(batches)
    cap = max(b[2] for b in batches) + BATCH  # always sufficient: tokens <= chars + 1 <= bytes + 1
    K, W, Q = args.steps, args.warmup, max(1, args.queue)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, min(args.streams, Q)))]
    ctxs = [DeviceContext(tok, streams[i % len(streams)].cuda_stream) for i in range(Q)]
    n_out = max(K, W, 1)
    out_tok = [torch.empty((cap, 6), dtype=torch.int32, device=dev) for _ in range(max(2 * Q, 24))]  # >= 2 gather chunks
    NB = len(out_tok)
    out_off_all = torch.empty((NB, BATCH + 1), dtype=torch.int64, device=dev)  # one row per output buffer
    out_off = [out_off_all[i] for i in range(NB)]
    out_st = [torch.empty(BATCH, dtype=torch.uint8, device=dev) for _ in range(len(out_tok))]

    def enqueue(i):
        d_utf8, d_off, total, _, _ = batches[i % nb]
        o = i % len(out_tok)
        ctxs[i % Q].tokenize(d_utf8.data_ptr(), d_off.data_ptr(), BATCH, total, out_tok[o].data_ptr(), cap,
                             out_off[o].data_ptr(), out_st[o].data_ptr())

    def drain():
        return [c.sync() for c in ctxs]

    # ---- untimed: device-side work counters of every distinct batch (algorithmic bytes)
    work = {k: 0 for k in ("sentences", "B", "C", "T", "N", "E", "K")}
    ctxs[0].set_profiling(PROFILE_WORK)
    for i in range(nb):
        d_utf8, d_off, total, _, _ = batches[i]
        ctxs[0].tokenize(d_utf8.data_ptr(), d_off.data_ptr(), BATCH, total, out_tok[0].data_ptr(), cap,
                         out_off[0].data_ptr(), out_st[0].data_ptr())
        ctxs[0].sync()
    for k, v in ctxs[0].work().items():
        work[k] += v
    ctxs[0].set_profiling(PROFILE_OFF)
    sample_tokens = None
    if rank == 0:  # keep batch 0's GPU result for the bit-exact check in the cpu_baseline leg
        d_utf8, d_off, total, _, _ = batches[0]
        ctxs[0].tokenize(d_utf8.data_ptr(), d_off.data_ptr(), BATCH, total, out_tok[0].data_ptr(), cap,
                         out_off[0].data_ptr(), out_st[0].data_ptr())
        nt = ctxs[0].sync()
        sample_tokens = (out_tok[0][:nt].cpu().numpy().copy(), out_off[0].cpu().numpy().copy())

    row_cache = {}
    size_pg = dist.new_group(backend="gloo") if multi else None  # CPU-side size exchange of the gather
    GC = 12  # steps per gather chunk (multi-rank): large enough that the host side of a gather (CPU-side size
             # exchange, one grouped send/recv call) is a small part of the chunk; the contexts' queue keeps the GPU fed

    def run_steps(nsteps):
        """Exactly `nsteps` steps; multi-rank: plus the overlapped gather of everything produced."""
        if not multi:
            for i in range(nsteps):
                enqueue(i)
            drain()
            return
        # Chunks of GC steps; the token records of chunk c travel to rank 0 (flat gatherv over xGMI) while
        # chunk c+1 is being tokenized.  Every byte produced is gathered.  Step i writes output buffer
        # i mod 2*GC, so chunk c's transfers are waited for before chunk c+2 starts.
        gather = ChunkedGather(dst=0, size_group=size_pg)
        ntok_of = {}
        chunks = []

        def retire(upto):  # steps < upto are complete on the device: collect their token counts
            for j in range(max(0, upto - Q), upto):
                if j not in ntok_of:
                    ntok_of[j] = ctxs[j % Q].sync()

        def post(lo, hi):
          t_post0 = time.perf_counter()
          if True:
            views = [out_tok[i % NB][: ntok_of[i]] for i in range(lo, hi)]  # sent as they are: no concat pass
            key = (lo % NB, hi - lo)
            if key not in row_cache:  # cached: no host-to-device copy (a sync behind a busy device) per chunk
                row_cache[key] = torch.tensor([i % NB for i in range(lo, hi)], device=dev)
            offs = out_off_all.index_select(0, row_cache[key])  # [steps, BATCH + 1] in one gather
            t_post1 = time.perf_counter()
            gather.post_steps(views, (offs[:, 1:] - offs[:, :-1]).reshape(-1), copy_own=False)  # rank 0's own records are already on rank 0
            if os.environ.get("BENCH_DEBUG_GATHER") and rank == 0:
                print(f"[gather] steps {lo}..{hi}: prepare {1e3 * (t_post1 - t_post0):.3f} ms, post {1e3 * (time.perf_counter() - t_post1):.3f} ms", file=sys.stderr)


        for c0 in range(0, nsteps, GC):
            if c0 >= 2 * GC:
                chunks += [(r[0].shape[0], sum(z[0] for z in r[2])) for r in gather.finish() if r is not None]  # chunk c0/GC - 2 has left its buffers
            t_l0 = time.perf_counter(); t_sync = 0.0
            for i in range(c0, min(c0 + GC, nsteps)):
                if i >= Q:
                    t_s0 = time.perf_counter()
                    ntok_of[i - Q] = ctxs[i % Q].sync()  # step i-Q used this ctx: done before it is reused
                    t_sync += time.perf_counter() - t_s0
                enqueue(i)
            if os.environ.get("BENCH_DEBUG_GATHER") and rank == 0:
                print(f"[loop] chunk at {c0}: {1e3 * (time.perf_counter() - t_l0):.3f} ms, of which waiting in sync {1e3 * t_sync:.3f} ms", file=sys.stderr)
            if c0 >= GC:
                retire(c0)
                post(c0 - GC, c0)
        last0 = ((nsteps - 1) // GC) * GC
        retire(nsteps)
        post(max(last0, 0), nsteps)
        chunks += [(r[0].shape[0], sum(z[0] for z in r[2])) for r in gather.finish() if r is not None]
        if rank == 0:
            assert chunks and all(got == want for got, want in chunks)

    if multi:  # untimed set-up of the gather path: index tensors of every chunk shape, allocator blocks of the chunk sizes
        for c0 in range(0, max(K, W, 1), GC):
            for total_steps in (K, W):
                n_st = min(c0 + GC, total_steps) - c0
                if n_st > 0 and (c0 % NB, n_st) not in row_cache:
                    row_cache[(c0 % NB, n_st)] = torch.tensor([i % NB for i in range(c0, c0 + n_st)], device=dev)
        if rank == 0:
            warm = [torch.empty((world * GC * cap // 3, 6), dtype=torch.int32, device=dev) for _ in range(2)]
            warm += [torch.empty(world * GC * BATCH, dtype=torch.int64, device=dev) for _ in range(2)]
            del warm  # stays in torch's caching allocator: no hipMalloc inside the timed region
        torch.cuda.synchronize()

    # ---- warmup (also brings up the RCCL channels of the gather)
    if W > 0:
        run_steps(W)
    for c in ctxs:
        if not os.environ.get("BENCH_NO_EVENTS"):
            c.set_profiling(PROFILE_EVENTS | PROFILE_SAMPLED)  # HIP events around every 4th launch
        c.profile(reset=True)

    # ---- timed region: exactly K steps
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(K)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = {"launches": 0, "tokenize_ms": 0.0, "aux_ms": 0.0}
    for c in ctxs:
        p = c.profile(reset=True)
        for k in prof:
            prof[k] += p[k]
        c.set_profiling(PROFILE_OFF)

    if rank != 0:
        dist.destroy_process_group()
        return

    sentences = K * BATCH * world
    bytes_in = sum(batches[i % nb][2] for i in range(K)) * world
    a, b, c_ = algorithmic_bytes(work)
    per_launch_bytes = (a + b + c_) / nb
    avg_kernel_s = prof["tokenize_ms"] / max(prof["launches"], 1) / 1e3
    achieved = per_launch_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from a separate rocprofv3 --pmc pass
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")  # rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, see file
        except Exception:
            traffic = None

    result = {
        "metric": "sentences/sec", "value": sentences / elapsed, "unit": "sentences/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: 100k synthetic ~40-char sentences per GPU, synthetic IPADIC-shaped "
                        "dictionary (392k records, 1316x1316 i16 matrix, 11 categories, 40 unk rows), batch=4096 "
                        "(the 24 full batches cycled), inputs resident in HBM, dense tokens left in HBM",
            "batch": BATCH, "sentences_per_gpu": N_SENT, "batches_in_flight": Q,
            "sharding": "sentences round-robin, dictionary replicated, one gatherv of token records to rank 0",
        },
        "input_MiB_per_s": bytes_in / elapsed / 2**20,
        "work_per_sentence": {k: work[k] / work["sentences"] for k in ("B", "C", "T", "N", "E", "K")},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": "k_tokenize_pool (fused lattice build + Viterbi + backtrace, LDS page pool)",
            "algorithmic_bytes_per_launch": per_launch_bytes,
            "stage_bytes_per_launch": {"A_lattice": a / nb, "B_viterbi": b / nb, "C_emit": c_ / nb},
            "avg_kernel_ms": avg_kernel_s * 1e3, "launches_timed": prof["launches"],
            "aux_kernels_avg_ms": prof["aux_ms"] / max(prof["launches"], 1),
            # `achieved` divides by the duration of ONE launch while `batches_in_flight` launches share the
            # chip (each therefore lasts ~that many times longer than its share of the work); the same
            # algorithmic bytes at the measured whole-job rate:
            "launches_in_flight": Q,
            "achieved_at_job_rate": per_launch_bytes * (sentences / world / BATCH) / elapsed / 1e9,
            "frac_at_job_rate": per_launch_bytes * (sentences / world / BATCH) / elapsed / 1e9 / HBM_PEAK_GBS,
        },
    }

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

    # ---- host-buffer entry point (H2D + kernels + D2H per batch): the PCIe-inclusive rate, never `value`
    if world == 1:
        from kanpyo_amd.tokenizer import TOKEN_DTYPE, pinned_empty
        h_out = (np.empty(cap, dtype=TOKEN_DTYPE), np.empty(BATCH + 1, dtype=np.uint64), np.empty(BATCH, dtype=np.uint8))
        tok.tokenize_packed(batches[0][3], batches[0][4], out=h_out)  # untimed: the pool ctx allocates its scratch, pages get touched
        t1 = time.perf_counter()
        done = 0
        for i in range(min(nb, 12)):
            _, _, _, utf8_h, offs_h = batches[i]
            tok.tokenize_packed(utf8_h, offs_h, out=h_out)
            done += BATCH
        result["pcie_inclusive"] = {"value": done / (time.perf_counter() - t1), "unit": "sentences/s",
                                    "what": "kgpu_tokenize_batch: pageable host buffers in, dense tokens out, one batch at a time"}
        # the same entry point given the 24 batches in ONE call (it pipelines 16384-sentence chunks over three
        # contexts), with pageable and with pinned (kgpu_host_alloc) buffers
        utf8_all, offs_all = pack_sentences(corpus[: nb * BATCH])
        capall = int(offs_all[-1]) // 2 + nb * BATCH  # tokens <= chars + 1 per sentence; the text is 3 bytes per char
        for name, alloc in (("large_call_pageable", np.empty), ("large_call_pinned", pinned_empty)):
            u = alloc(utf8_all.shape, dtype=np.uint8); u[:] = utf8_all
            o = alloc(offs_all.shape, dtype=np.uint64); o[:] = offs_all
            big = (alloc(capall, dtype=TOKEN_DTYPE), alloc(nb * BATCH + 1, dtype=np.uint64), alloc(nb * BATCH, dtype=np.uint8))
            tok.tokenize_packed(u, o, out=big)  # untimed: scratch allocation, pages touched
            t1 = time.perf_counter()
            for _ in range(3):
                tok.tokenize_packed(u, o, out=big)
            result["pcie_inclusive"][name] = 3 * nb * BATCH / (time.perf_counter() - t1)
            del big

    # ---- CPU baseline (rank 0, N==1 only): the oracle restatement on the host cores
    if world == 1 and not args.no_cpu:
        from oracle import oracle

        orc = oracle.OracleTokenizer.from_dict(sd.dict)
        utf8, offs = pack_sentences(corpus)
        done, t_cpu, exp0 = 0, 0.0, None
        while t_cpu < args.cpu_seconds:
            t1 = time.perf_counter()
            r = orc.tokenize_batch(utf8, offs, 1)
            t_cpu += time.perf_counter() - t1
            done += len(corpus)
            exp0 = r
        ncores = os.cpu_count() or 1
        t1 = time.perf_counter()
        orc.tokenize_batch(utf8, offs, ncores)
        t_all = time.perf_counter() - t1
        # bit-exact check of the GPU's batch 0 against the same sentences from the oracle
        n0 = int(exp0.offsets[BATCH])
        g_tok, g_off = sample_tokens
        exact = bool(np.array_equal(g_off.astype(np.uint64), exp0.offsets[: BATCH + 1])
                     and np.array_equal(g_tok.reshape(-1), exp0.tokens[:n0].view(np.int32).reshape(-1).astype(np.int32)))
        result["cpu_baseline"] = {
            "value": done / t_cpu, "unit": "sentences/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"the same 100k-sentence cfg2 corpus, {done // len(corpus)} pass(es), {t_cpu:.1f} s, "
                      "oracle/kanpyo_oracle.c (CPU restatement of Kanpyo's algorithm, gcc -O2), single thread",
            "all_cores": {"value": len(corpus) / t_all, "cores": ncores},
            "gpu_batch0_bit_exact": exact,
        }
        result["speedup_vs_cpu_1thread"] = result["value"] / result["cpu_baseline"]["value"]
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(result), flush=True)
    if multi:
        os.dup2(2, 1)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

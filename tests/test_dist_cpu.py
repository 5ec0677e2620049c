"""Multi-GPU path on CPU: world_size-2 gloo run of the round-robin sharding and
the flat gatherv of token records (kanpyo_amd/dist.py).  The per-rank tokens come
from the oracle (no GPU here); the gathered + reassembled stream must equal the
oracle's stream over the unsharded corpus."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from kanpyo_amd import synth
    from kanpyo_amd.dist import gather_tokens, reassemble, shard_indices
    from kanpyo_amd.tokenizer import pack_sentences
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.build_dict(6000, seed=5)
    corpus = synth.make_corpus(sd, 301, 3, "cfg2") + ["", "あ"]  # odd count: ragged shards
    orc = oracle.OracleTokenizer.from_dict(sd.dict)
    mine = shard_indices(len(corpus), rank, world)
    utf8, offs = pack_sentences([corpus[i] for i in mine])
    r = orc.tokenize_batch(utf8, offs, 1)
    tok = torch.from_numpy(r.tokens.view(np.int32).reshape(-1, 6).copy())
    cnt = torch.from_numpy((r.offsets[1:] - r.offsets[:-1]).astype(np.int64))
    tok_all, cnt_all, sizes = gather_tokens(tok, cnt, dst=0)
    if rank == 0:
        assert [s[1] for s in sizes] == [len(shard_indices(len(corpus), k, world)) for k in range(world)]
        got_t, got_off = reassemble(tok_all.numpy(), cnt_all.numpy(), len(corpus), world)
        full = orc.tokenize_batch(*pack_sentences(corpus), 1)
        ok = np.array_equal(got_off.astype(np.uint64), full.offsets) and np.array_equal(
            got_t.reshape(-1), full.tokens.view(np.int32).reshape(-1))
        open(os.path.join(tmpdir, "result"), "w").write("ok" if ok else "mismatch")
    else:
        assert tok_all is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_round_robin_shard_and_gather_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "result").read() == "ok"


def test_unshard_order_is_a_permutation():
    from kanpyo_amd.dist import shard_indices, unshard_order

    for n, w in ((0, 2), (1, 2), (7, 2), (8, 8), (100, 3)):
        perm = unshard_order(n, w)
        gathered = np.concatenate([shard_indices(n, r, w) for r in range(w)]) if n else np.zeros(0, dtype=np.int64)
        assert np.array_equal(gathered[perm], np.arange(n))


def _worker_chunked(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from kanpyo_amd.dist import ChunkedGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = ChunkedGather(dst=0)
    sent = []
    for c in range(4):  # ragged chunks, one of them empty on rank 1
        nt = 0 if (rank == 1 and c == 2) else 10 * (rank + 1) + c
        tok = (torch.arange(nt * 6, dtype=torch.int32).reshape(nt, 6) + 1000 * rank + 100000 * c)
        cnt = torch.full((3 + rank,), c, dtype=torch.int64)
        sent.append((tok, cnt))
        g.post(tok, cnt)
    res = g.finish()
    if rank == 0:
        ok = len(res) == 4
        for c, (tok_all, cnt_all, sizes) in enumerate(res):
            t0 = 0
            for r, (nt, nc) in enumerate(sizes):
                exp_nt = 0 if (r == 1 and c == 2) else 10 * (r + 1) + c
                exp = torch.arange(exp_nt * 6, dtype=torch.int32).reshape(exp_nt, 6) + 1000 * r + 100000 * c
                ok = ok and nt == exp_nt and torch.equal(tok_all[t0 : t0 + nt], exp) and nc == 3 + r
                t0 += nt
            ok = ok and cnt_all.tolist() == sum(([c] * (3 + r) for r in range(world)), [])
        open(os.path.join(tmpdir, "chunked"), "w").write("ok" if ok else "mismatch")
    else:
        assert all(r is None for r in res)
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_overlapped_gather_gloo(tmp_path):
    import torch.multiprocessing as mp

    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_chunked, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert open(tmp_path / "chunked").read() == "ok"


def _worker_steps(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from kanpyo_amd.dist import ChunkedGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for size_group in (None, dist.new_group(backend="gloo")):
        g = ChunkedGather(dst=0, size_group=size_group)
        for c in range(3):  # chunks of 4 steps each, ragged, with empty steps
            views = []
            for j in range(4):
                nt = 0 if (rank + j + c) % 4 == 0 else 3 * (rank + 1) + j
                views.append(torch.full((nt, 6), 100 * c + 10 * j + rank, dtype=torch.int32))
            g.post_steps(views, torch.full((2 + rank,), c, dtype=torch.int64), copy_own=size_group is None)
        res = g.finish()
        if rank == 0:
            for c, r_ in enumerate(res):
                tok_all, cnt_all, sizes = r_[0], r_[1], r_[2]
                if len(r_) == 4:  # copy_own=False: the root's slice is left to the caller
                    torch.cat([v for v in r_[3] if v.shape[0]], out=tok_all[: sizes[0][0]]) if sizes[0][0] else None
                exp = []
                for r in range(world):
                    for j in range(4):
                        nt = 0 if (r + j + c) % 4 == 0 else 3 * (r + 1) + j
                        exp.append(torch.full((nt, 6), 100 * c + 10 * j + r, dtype=torch.int32))
                ok = ok and torch.equal(tok_all, torch.cat(exp)) and cnt_all.tolist() == sum(([c] * (2 + r) for r in range(world)), [])
                ok = ok and sizes == [(sum(0 if (r + j + c) % 4 == 0 else 3 * (r + 1) + j for j in range(4)), 2 + r) for r in range(world)]
        else:
            ok = ok and all(r is None for r in res)
    if rank == 0:
        open(os.path.join(tmpdir, "steps"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_stepwise_gather_without_concat_gloo(tmp_path):
    """ChunkedGather.post_steps: every step's records sent as they are, received in place on the root."""
    import torch.multiprocessing as mp

    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_steps, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert open(tmp_path / "steps").read() == "ok"


class _OracleEngine:
    """CPU stand-in for bench.GpuEngine (same protocol): tokens come from the oracle, tensors live on the CPU."""

    def __init__(self, orc, wl):
        self.orc, self.wl, self.done = orc, wl, {}

    def nb(self, step):
        return self.wl.nb(step)

    def enqueue(self, step, b):
        utf8, offs = self.wl.packed[step % len(self.wl.packed)][b]
        self.done[(step, b)] = self.orc.tokenize_batch(utf8, offs, 1)

    def results(self, step):
        import torch

        views, counts = [], []
        for b in range(self.nb(step)):
            r = self.done.pop((step, b))
            views.append(torch.from_numpy(r.tokens.view(np.int32).reshape(-1, 6).copy()))
            counts.append(torch.from_numpy((r.offsets[1:] - r.offsets[:-1]).astype(np.int64)))
        return views, torch.cat(counts)

    def after_gather(self):
        pass

    def drain(self):
        assert not self.done


def _worker_bench_job(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import bench
    from kanpyo_amd import synth
    from kanpyo_amd.dist import ChunkedGather, reassemble
    from kanpyo_amd.tokenizer import pack_sentences
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.build_dict(6000, seed=5)
    corpora = [synth.make_corpus(sd, 301 + 7 * k, 100 + k, "cfg2") + ([""] if k == 1 else []) for k in range(3)]  # ragged shards and batches
    orc = oracle.OracleTokenizer.from_dict(sd.dict)
    wl = bench.Workload(corpora, rank, world, batch=64)
    ok = True
    for chunk_steps, nsteps in ((1, 4), (2, 5), (3, 2)):
        chunks = {}
        bench.run_job(_OracleEngine(orc, wl), nsteps, ChunkedGather(dst=0), chunk_steps, lambda c0, r: chunks.__setitem__(c0, r))
        ok = ok and sorted(chunks) == list(range(0, nsteps, chunk_steps))
        if rank != 0:
            ok = ok and all(r is None for r in chunks.values())
            continue
        for c0, (tok_all, cnt_all, sizes) in chunks.items():
            steps = range(c0, min(c0 + chunk_steps, nsteps))
            if chunk_steps == 1:  # one step per chunk: the reassembled stream is the unsharded corpus's stream
                n = len(corpora[c0 % 3])
                got_t, got_off = reassemble(tok_all.numpy(), cnt_all.numpy(), n, world)
                full = orc.tokenize_batch(*pack_sentences(corpora[c0 % 3]), 1)
                ok = ok and np.array_equal(got_off.astype(np.uint64), full.offsets)
                ok = ok and np.array_equal(got_t.reshape(-1), full.tokens.view(np.int32).reshape(-1))
            exp_t, exp_c = [], []  # rank-major: every rank's steps of the chunk, in order
            for r in range(world):
                for s in steps:
                    c = corpora[s % 3]
                    e = orc.tokenize_batch(*pack_sentences([c[i] for i in range(r, len(c), world)]), 1)
                    exp_t.append(e.tokens.view(np.int32).reshape(-1, 6))
                    exp_c.append((e.offsets[1:] - e.offsets[:-1]).astype(np.int64))
            ok = ok and np.array_equal(tok_all.numpy(), np.concatenate(exp_t)) and np.array_equal(cnt_all.numpy(), np.concatenate(exp_c))
    if rank == 0:
        open(os.path.join(tmpdir, "benchjob"), "w").write("ok" if ok else "mismatch")
    else:
        assert ok
    dist.barrier()
    dist.destroy_process_group()


def test_bench_job_shards_and_gathers_gloo(tmp_path):
    """bench.py's own run_job / Workload (cfg 4: sentence i -> rank i mod G, chunked gather to rank 0) with world
    size 2 over gloo and oracle-produced tokens: the gathered stream of every chunk is the rank-major
    concatenation, and a one-step chunk reassembles to the unsharded corpus's token stream."""
    import torch.multiprocessing as mp

    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_worker_bench_job, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "benchjob").read() == "ok"


def test_bench_chunking_rule():
    import bench

    assert [bench.chunk_steps_for(nb) for nb in (25, 13, 7, 4, 1)] == [1, 1, 2, 3, 12]


class _CompactOracleEngine(_OracleEngine):
    """The same stand-in with bench.GpuEngine(compact=True)'s protocol: 8-byte kgpu_token8 rows [k, 2] and, behind the counts,
    the first token's (position, start) of every sentence as one int64 each (include/kanpyo_gpu.h: kgpu_token8)."""

    def results(self, step):
        import torch

        views, counts, firsts = [], [], []
        for b in range(self.nb(step)):
            r = self.done.pop((step, b))
            t = r.tokens
            packed = (t["cls"] | ((t["end"] - t["start"]) << 2) | (t["byte_len"] << 14)).astype(np.uint32)
            t8 = np.stack([t["id"].astype(np.int32).view(np.uint32), packed], axis=1).view(np.int32)
            cnt = (r.offsets[1:] - r.offsets[:-1]).astype(np.int64)
            first = np.full((len(cnt), 2), 0xFFFFFFFF, dtype=np.uint32)
            has = cnt > 0
            at = r.offsets[:-1][has].astype(np.int64)
            first[has, 0], first[has, 1] = t["position"][at], t["start"][at]
            views.append(torch.from_numpy(np.ascontiguousarray(t8)))
            counts.append(torch.from_numpy(cnt))
            firsts.append(torch.from_numpy(first.view(np.int64).reshape(-1).copy()))
        return views, torch.cat(counts + firsts) if counts else torch.zeros(0, dtype=torch.int64)


def _worker_compact_job(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import bench
    from kanpyo_amd import synth
    from kanpyo_amd.dist import ChunkedGather, reassemble
    from kanpyo_amd.tokenizer import pack_sentences
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.build_dict(6000, seed=5)
    corpora = [synth.make_corpus(sd, 300 + 11 * k, 100 + k, "cfg2") + (["", "あ"] if k == 0 else []) for k in range(2)]
    orc = oracle.OracleTokenizer.from_dict(sd.dict)
    # one batch per step: the engine's layout [counts of the step | firsts of the step] is what bench.expand_gathered slices
    wl = bench.Workload(corpora, rank, world, batch=4096)
    ok = True
    for chunk_steps, nsteps in ((1, 2), (2, 4)):
        chunks = {}
        bench.run_job(_CompactOracleEngine(orc, wl), nsteps, ChunkedGather(dst=0), chunk_steps, lambda c0, r: chunks.__setitem__(c0, r))
        if rank != 0:
            continue
        for c0, (tok8_all, cnt2_all, sizes) in chunks.items():
            steps = list(range(c0, min(c0 + chunk_steps, nsteps)))
            per_rank = [[(len(corpora[s % 2]) - r + world - 1) // world for s in steps] for r in range(world)]
            tok, cnt = bench.expand_gathered(tok8_all.numpy(), cnt2_all.numpy(), sizes, per_rank)
            exp_t, exp_c = [], []
            for r in range(world):
                for s in steps:
                    c = corpora[s % 2]
                    e = orc.tokenize_batch(*pack_sentences([c[i] for i in range(r, len(c), world)]), 1)
                    exp_t.append(e.tokens.view(np.int32).reshape(-1, 6))
                    exp_c.append((e.offsets[1:] - e.offsets[:-1]).astype(np.int64))
            ok = ok and np.array_equal(tok, np.concatenate(exp_t)) and np.array_equal(cnt, np.concatenate(exp_c))
            if chunk_steps == 1:
                n = len(corpora[c0 % 2])
                got_t, got_off = reassemble(tok, cnt, n, world)
                full = orc.tokenize_batch(*pack_sentences(corpora[c0 % 2]), 1)
                ok = ok and np.array_equal(got_off.astype(np.uint64), full.offsets) and np.array_equal(got_t.reshape(-1), full.tokens.view(np.int32).reshape(-1))
    if rank == 0:
        open(os.path.join(tmpdir, "compactjob"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_bench_job_gathers_compact_records_gloo(tmp_path):
    """bench.py's default wire format for N > 1: 8-byte kgpu_token8 records + firsts behind the counts, gathered chunk by chunk with
    world size 2 over gloo, expanded on the root by bench.expand_gathered (kgpu_expand_tokens: a host function of the library, no GPU
    needed) -- the expanded stream equals the oracle's 24-byte records, rank-major, and reassembles to the unsharded corpus's stream."""
    import torch.multiprocessing as mp

    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_worker_compact_job, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "compactjob").read() == "ok"

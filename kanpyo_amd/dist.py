"""Multi-GPU sharding and the result gather.

Sentences are independent (Tokenizer::tokenize takes &self and a single &str,
reference src/tokenizer.rs:16), so the path shards with no data-path exchange:
sentence i goes to rank i mod G, the dictionary is replicated per GPU.  The only
communication is one variable-length gather of the 24-byte token records to
rank 0: every peer has its own direct xGMI link to the root, so a flat
gatherv (grouped send/recv, RCCL's ncclSend/ncclRecv under torch.distributed's
"nccl" backend) uses all links in parallel -- no ring, no all-reduce.
A sentence is never split across GPUs (Viterbi is serial along the sentence).
torch is imported lazily: it is plumbing (device memory, process group), not
part of the tokenizer.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def shard_indices(n: int, rank: int, world: int) -> np.ndarray:
    """Round-robin: the sentences rank `rank` owns."""
    return np.arange(rank, n, world, dtype=np.int64)


def unshard_order(n: int, world: int) -> np.ndarray:
    """Position in the rank-major gathered stream of each original sentence:
    gathered = [rank0's sentences..., rank1's..., ...]; returns perm with
    gathered[perm[i]] == sentence i."""
    sizes = [(n - r + world - 1) // world for r in range(world)]
    starts = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    i = np.arange(n, dtype=np.int64)
    return starts[i % world] + i // world


def gather_tokens(tokens, counts, dst: int = 0, group=None):
    """Flat gatherv of token records and per-sentence counts to rank `dst`.

    tokens: [T, 6] int32 tensor (kgpu_token rows) on this rank's device (or CPU
    for the gloo tests); counts: [n_local] int64 tokens per local sentence.
    Returns (tokens_all, counts_all, sizes) on dst -- rank-major concatenation --
    and (None, None, sizes) elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = torch.tensor([tokens.shape[0], counts.shape[0]], dtype=torch.int64, device=tokens.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    sizes = [(int(m[0]), int(m[1])) for m in metas]
    if rank == dst:
        tok_all = torch.empty((sum(s[0] for s in sizes), 6), dtype=tokens.dtype, device=tokens.device)
        cnt_all = torch.empty(sum(s[1] for s in sizes), dtype=counts.dtype, device=counts.device)
        ops, t0, c0 = [], 0, 0
        for r, (nt, nc) in enumerate(sizes):
            tv, cv = tok_all[t0 : t0 + nt], cnt_all[c0 : c0 + nc]
            if r == dst:
                tv.copy_(tokens)
                cv.copy_(counts)
            else:
                if nt:
                    ops.append(dist.P2POp(dist.irecv, tv, r, group))
                if nc:
                    ops.append(dist.P2POp(dist.irecv, cv, r, group))
            t0 += nt
            c0 += nc
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return tok_all, cnt_all, sizes
    ops = []
    if tokens.shape[0]:
        ops.append(dist.P2POp(dist.isend, tokens.contiguous(), dst, group))
    if counts.shape[0]:
        ops.append(dist.P2POp(dist.isend, counts.contiguous(), dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None, None, sizes


def _exchange_sizes(n_tok: int, n_cnt: int, device, world: int, group, size_group=None):
    """Every rank's (token rows, count entries).  With `size_group` (a CPU/gloo group over the same ranks)
    the exchange never touches the GPU -- no wait behind the kernels queued on a busy device; otherwise
    one small all-gather on `group` and ONE device-to-host read."""
    import torch
    import torch.distributed as dist

    if size_group is not None:
        meta = torch.tensor([n_tok, n_cnt], dtype=torch.int64)
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta, group=size_group)
        return [(int(m[0]), int(m[1])) for m in metas]
    meta = torch.tensor([n_tok, n_cnt], dtype=torch.int64, device=device)
    allm = torch.empty((world, 2), dtype=torch.int64, device=device)
    try:
        dist.all_gather_into_tensor(allm, meta, group=group)
    except (RuntimeError, NotImplementedError):  # backend without the flat form
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta, group=group)
        allm = torch.stack(metas)
    return [(int(a), int(b)) for a, b in allm.cpu().tolist()]


class ChunkedGather:
    """The same flat gatherv, posted chunk by chunk so that it overlaps the tokenization of
    the following chunk (the root's seven inbound xGMI links work while the CUs compute).

    Every rank calls post() the same number of times in the same order; finish() waits for
    all transfers and, on `dst`, returns the rank-major (tokens, counts, sizes) of each chunk.
    """

    def __init__(self, dst: int = 0, group=None, size_group=None):
        self.dst, self.group, self.size_group = dst, group, size_group
        self._inflight = []  # (works, result-or-None, keepalive)

    def post(self, tokens, counts):
        import torch
        import torch.distributed as dist

        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        sizes = _exchange_sizes(tokens.shape[0], counts.shape[0], tokens.device, world, self.group, self.size_group)
        ops, result = [], None
        tokens, counts = tokens.contiguous(), counts.contiguous()
        if rank == self.dst:
            tok_all = torch.empty((sum(s[0] for s in sizes), 6), dtype=tokens.dtype, device=tokens.device)
            cnt_all = torch.empty(sum(s[1] for s in sizes), dtype=counts.dtype, device=counts.device)
            t0 = c0 = 0
            for r, (nt, nc) in enumerate(sizes):
                tv, cv = tok_all[t0 : t0 + nt], cnt_all[c0 : c0 + nc]
                if r == rank:
                    tv.copy_(tokens)
                    cv.copy_(counts)
                else:
                    if nt:
                        ops.append(dist.P2POp(dist.irecv, tv, r, self.group))
                    if nc:
                        ops.append(dist.P2POp(dist.irecv, cv, r, self.group))
                t0 += nt
                c0 += nc
            result = (tok_all, cnt_all, sizes)
        else:
            if tokens.shape[0]:
                ops.append(dist.P2POp(dist.isend, tokens, self.dst, self.group))
            if counts.shape[0]:
                ops.append(dist.P2POp(dist.isend, counts, self.dst, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        self._inflight.append((works, result, (tokens, counts)))

    def post_steps(self, token_views, counts, copy_own: bool = True):
        """Same gather for a chunk made of several steps whose token records live in separate buffers:
        no concatenation on the senders (each view is sent as it is, all sends of the chunk in one
        grouped call), the root receives every view straight into its place of the rank-major
        stream and copies its own views there in one pass.  `token_views`: list of [k_i, 6] tensors
        (the same number of steps on every rank); `counts`: [n_local] tokens per local sentence.
        copy_own=False: the root's own records are already where the gather wants them (on the root), so
        they are left in their buffers -- the root's slice of the returned stream is then uninitialised and
        its views are returned as a fourth element instead."""
        import torch
        import torch.distributed as dist

        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        mine = [int(v.shape[0]) for v in token_views] + [int(counts.shape[0])]
        if self.size_group is not None:
            meta = torch.tensor(mine, dtype=torch.int64)
            metas = [torch.zeros_like(meta) for _ in range(world)]
            dist.all_gather(metas, meta, group=self.size_group)
            table = [m.tolist() for m in metas]
        else:
            meta = torch.tensor(mine, dtype=torch.int64, device=counts.device)
            metas = [torch.zeros_like(meta) for _ in range(world)]
            dist.all_gather(metas, meta, group=self.group)
            table = torch.stack(metas).cpu().tolist()
        sizes = [(sum(row[:-1]), row[-1]) for row in table]
        ops, result = [], None
        counts = counts.contiguous()
        if rank == self.dst:
            dev = counts.device
            width = int(token_views[0].shape[1]) if token_views else 6  # 6: kgpu_token rows, 2: kgpu_token8 rows
            tok_all = torch.empty((sum(s[0] for s in sizes), width), dtype=token_views[0].dtype if token_views else torch.int32, device=dev)
            cnt_all = torch.empty(sum(s[1] for s in sizes), dtype=counts.dtype, device=dev)
            t0 = c0 = 0
            for r, row in enumerate(table):
                nt, nc = sizes[r]
                if r == rank:
                    if nt and copy_own:
                        torch.cat([v for v in token_views if v.shape[0]], out=tok_all[t0 : t0 + nt])
                    cnt_all[c0 : c0 + nc].copy_(counts)
                else:
                    at = t0
                    for k in row[:-1]:
                        if k:
                            ops.append(dist.P2POp(dist.irecv, tok_all[at : at + k], r, self.group))
                        at += k
                    if nc:
                        ops.append(dist.P2POp(dist.irecv, cnt_all[c0 : c0 + nc], r, self.group))
                t0 += nt
                c0 += nc
            result = (tok_all, cnt_all, sizes) if copy_own else (tok_all, cnt_all, sizes, list(token_views))
        else:
            for v in token_views:
                if v.shape[0]:
                    ops.append(dist.P2POp(dist.isend, v, self.dst, self.group))
            if counts.shape[0]:
                ops.append(dist.P2POp(dist.isend, counts, self.dst, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        self._inflight.append((works, result, (token_views, counts)))

    def finish(self):
        out = []
        for works, result, _keep in self._inflight:
            for w in works:
                w.wait()
            out.append(result)
        self._inflight = []
        return out


def reassemble(tok_all: np.ndarray, cnt_all: np.ndarray, n: int, world: int):
    """Rank-major gathered stream -> original sentence order (host side).
    Returns (tokens [T,6], tok_offsets [n+1])."""
    perm = unshard_order(n, world)
    off_g = np.concatenate([[0], np.cumsum(cnt_all)]).astype(np.int64)
    cnt = cnt_all[perm]
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    out = np.empty_like(tok_all)
    src_start = off_g[perm]
    # vectorised segment copy
    idx = np.repeat(src_start - off[:-1], cnt) + np.arange(off[-1])
    out[:] = tok_all[idx]
    return out, off

"""The host-side merge of kgpu_tokenize_batch_multi (kgpu_multi.cpp: merge_plan + merge_slice on the worker pool), driven WITHOUT a device through
kgpu_debug_merge_shards: G shard blocks as the shards' compaction kernels leave them in mapped host memory (8-byte records, first (position, start)
per sentence, local token offsets, status bytes) -> the caller's order (sentence j of the super-chunk = shard j mod G, local index j / G), 24-byte
records, global offsets.  Compared with a plain Python / numpy restatement; the G = 8 case is also timed (the reference's Tokenizer::tokenize is &self
with no cross-call state, src/tokenizer.rs:16 -- sentences shard freely; the merge is the only serial step of the multi-device call)."""
import json
import os

import numpy as np
import pytest

from kanpyo_amd import _lib
from kanpyo_amd.tokenizer import TOKEN_DTYPE, merge_bench, merge_shards

T8 = np.dtype([("id", "<i4"), ("packed", "<u4")])


def fabricate(G, cnt, rng, max_tok=48, empty_rate=0.05):
    """Shard blocks of one super-chunk of cnt sentences, and the expected merged result."""
    shards = []
    for g in range(G):
        m = (cnt - g + G - 1) // G if cnt > g else 0
        k = rng.integers(1, max_tok, size=m)
        k[rng.random(m) < empty_rate] = 0  # an unreachable EOS gives an empty token list (lattice.rs:144-153)
        toff = np.concatenate([[0], np.cumsum(k)]).astype(np.uint64)
        nt = int(toff[-1])
        chars = rng.integers(1, 12, size=nt).astype(np.uint32)
        nbytes = chars * rng.integers(1, 4, size=nt).astype(np.uint32)
        cls = rng.integers(0, 3, size=nt).astype(np.uint32)
        rec = np.zeros(nt, dtype=T8)
        rec["id"] = rng.integers(0, 400000, size=nt)
        rec["packed"] = cls | (chars << 2) | (nbytes << 14)
        first = rng.integers(0, 5, size=(m, 2)).astype(np.uint32)
        st = rng.integers(0, 2, size=m).astype(np.uint8)
        shards.append(dict(m=m, toff=toff, rec=rec, first=first, st=st, chars=chars, nbytes=nbytes, cls=cls))
    return shards


def expected(shards, G, cnt):
    counts = np.zeros(cnt, dtype=np.uint64)
    for g, sh in enumerate(shards):
        counts[g::G] = np.diff(sh["toff"])
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    out = np.zeros(int(offs[-1]), dtype=TOKEN_DTYPE)
    status = np.zeros(cnt, dtype=np.uint8)
    for g, sh in enumerate(shards):
        status[g::G] = sh["st"]
        for k in range(sh["m"]):
            t0, t1 = int(sh["toff"][k]), int(sh["toff"][k + 1])
            if t1 == t0:
                continue
            j = g + k * G
            o = int(offs[j])
            ch, by = sh["chars"][t0:t1].astype(np.uint64), sh["nbytes"][t0:t1].astype(np.uint64)
            start = int(sh["first"][k, 1]) + np.concatenate([[0], np.cumsum(ch)[:-1]])
            pos = int(sh["first"][k, 0]) + np.concatenate([[0], np.cumsum(by)[:-1]])
            seg = out[o:o + t1 - t0]
            seg["id"] = sh["rec"]["id"][t0:t1]
            seg["cls"] = sh["cls"][t0:t1]
            seg["position"] = pos
            seg["start"] = start
            seg["end"] = start + ch
            seg["byte_len"] = by
    return out, offs, status


def run_merge(shards, G, cnt, slice_, reps=1, capacity=None, want_tokens=True):
    return merge_shards([(sh["rec"], sh["first"], sh["toff"], sh["st"]) for sh in shards], cnt, slice_, reps, capacity, want_tokens)


@pytest.mark.parametrize("G,cnt,slice_", [(1, 100, 2048), (2, 1, 2048), (2, 4097, 2048), (3, 5000, 64), (5, 3, 2048), (7, 12345, 1000), (8, 65536, 2048), (8, 7, 2), (64, 1000, 33)])
def test_merge_equals_plain_restatement(G, cnt, slice_):
    rng = np.random.default_rng(G * 1000 + cnt)
    shards = fabricate(G, cnt, rng)
    exp_tok, exp_off, exp_st = expected(shards, G, cnt)
    rc, tok, off, st, n_tok, _ = run_merge(shards, G, cnt, slice_)
    assert rc == 0, _lib.lib().kgpu_last_error()
    assert n_tok == int(exp_off[-1])
    assert np.array_equal(off, exp_off)
    assert np.array_equal(st, exp_st)
    assert np.array_equal(tok, exp_tok)


def test_merge_capacity_and_offsets_only():
    G, cnt = 4, 3000
    shards = fabricate(G, cnt, np.random.default_rng(7))
    _, exp_off, exp_st = expected(shards, G, cnt)
    rc, _, off, st, n_tok, _ = run_merge(shards, G, cnt, 512, capacity=10)
    assert rc == _lib.KGPU_ERR_CAPACITY and n_tok == int(exp_off[-1])       # the needed size is reported, the offsets and status bytes are still written
    assert np.array_equal(off, exp_off) and np.array_equal(st, exp_st)
    rc, _, off, st, n_tok, _ = run_merge(shards, G, cnt, 512, want_tokens=False)
    assert rc == 0 and np.array_equal(off, exp_off) and np.array_equal(st, exp_st)


@pytest.mark.parametrize("G,cnt,slice_", [(1, 100, 2048), (2, 4097, 2048), (3, 5000, 64), (8, 65536, 2048), (8, 7, 2), (64, 1000, 33)])
def test_compact_merge_expands_to_the_same_records(G, cnt, slice_):
    """kgpu_tokenize_batch_multi_compact's merge: the shards' 8-byte records + firsts in the caller's order; kgpu_expand_tokens over the merged arrays gives
    exactly the 24-byte records of the other form (and the plain restatement)."""
    from kanpyo_amd.device import expand_tokens

    rng = np.random.default_rng(G * 7919 + cnt)
    shards = fabricate(G, cnt, rng)
    exp_tok, exp_off, exp_st = expected(shards, G, cnt)
    rc, (tok8, first), off, st, n_tok, _ = merge_shards([(sh["rec"], sh["first"], sh["toff"], sh["st"]) for sh in shards], cnt, slice_, compact=True)
    assert rc == 0, _lib.lib().kgpu_last_error()
    assert n_tok == int(exp_off[-1]) and len(tok8) == n_tok and tok8.itemsize == 8
    assert np.array_equal(off, exp_off) and np.array_equal(st, exp_st)
    for g, sh in enumerate(shards):   # the firsts travel with their sentences
        assert np.array_equal(first[g::G], sh["first"])
    assert np.array_equal(expand_tokens(tok8, off, first), exp_tok)
    rc, _, off2, st2, n2, _ = merge_shards([(sh["rec"], sh["first"], sh["toff"], sh["st"]) for sh in shards], cnt, slice_, token_capacity=1, compact=True)
    if n_tok > 1:
        assert rc == _lib.KGPU_ERR_CAPACITY and n2 == n_tok and np.array_equal(off2, exp_off) and np.array_equal(st2, exp_st)


def test_merge_throughput_eight_shards():
    """G = 8, a super-chunk of 8 x 8192 sentences with cfg 2's ~32 tokens each: the rate of the merge alone on this box's CPUs (bench.py reports the
    same measurement as `multi_merge`); the assertion is only a floor far below any healthy box -- the 24-byte expansion is a memory-bandwidth job."""
    r = merge_bench(8, 8192, 32, reps=10)
    print(json.dumps(r))
    assert r["sentences_per_s"] > 3e6
    rc = merge_bench(8, 8192, 32, reps=10, compact=True)   # the 8-byte-record form moves a third of the bytes
    print(json.dumps(rc))
    assert rc["record_bytes"] == 8 and rc["sentences_per_s"] > 3e6


@pytest.mark.parametrize("n,misalign", [(3, 0), (500, 0), (5000, 0), (5000, 4)])
def test_expand_tokens_small_and_streamed(n, misalign):
    """kgpu_expand_tokens (the public host-side expansion of kgpu_token8 records): below 32 768 tokens ordinary stores, above non-temporal ones (8-byte
    aligned output only) -- the same records either way, also into an output that is only 4-byte aligned."""
    import ctypes as C

    rng = np.random.default_rng(n + misalign)
    sh = fabricate(1, n, rng, max_tok=24, empty_rate=0.1)[0]
    exp_tok, exp_off, _ = expected([sh], 1, n)
    buf = np.zeros(len(exp_tok) * 24 + 64, dtype=np.uint8)
    base = (-buf.ctypes.data) % 8 + misalign
    out = buf[base:base + len(exp_tok) * 24]
    L = _lib.lib()
    first = np.ascontiguousarray(sh["first"].reshape(-1))
    L.kgpu_expand_tokens(sh["rec"].ctypes.data, sh["toff"].ctypes.data, first.ctypes.data, n, out.ctypes.data)
    assert np.array_equal(out.view(np.uint8), exp_tok.view(np.uint8).reshape(-1))

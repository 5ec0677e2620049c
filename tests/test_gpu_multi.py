"""The multi-device entry points of the C ABI on ONE GPU: a device list [0, 0, 0] exercises the shard (sentence i -> entry i mod G), the
per-entry contexts and threads, and the reassembly in the caller's original order; the result must be byte for byte the single-device one and
the oracle's.  (Reference src/tokenizer.rs:16: &self, Send + Sync -- sentences are independent, so sharding changes nothing but the order of
work; BASELINE cfg 4.)  On a node with several GPUs the same code runs with one handle per device; only the device ordinals differ."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from kanpyo_amd import Tokenizer, _lib, synth
    from oracle import oracle

    assert _lib.lib().kgpu_device_count() > 0
    oracle.build()
    sd = synth.build_dict(20000, seed=11)
    toks = [Tokenizer(sd.dict, device=0) for _ in range(3)]  # three handles (three dictionary copies) on device 0
    return sd, toks, oracle.OracleTokenizer.from_dict(sd.dict)


def _check(toks, orc, utf8, offs, **kw):
    from kanpyo_amd.tokenizer import tokenize_packed_multi

    t, toff, st = tokenize_packed_multi(toks, utf8, offs, **kw)
    exp = orc.tokenize_batch(utf8, offs, 8)
    assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens)
    return t, toff, st


@pytest.mark.parametrize("G", [1, 2, 3])
def test_sharded_call_equals_one_device_and_oracle(env, G, monkeypatch):
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, toks, orc = env
    sents = synth.make_corpus(sd, 6000, 3, "cfg2") + synth.make_corpus(sd, 300, 4, "cfg3") + ["", "すもももももももものうち", "テ"]
    rng = np.random.default_rng(G)
    rng.shuffle(sents)
    utf8, offs = pack_sentences(sents)
    t, toff, st = _check(toks[:G], orc, utf8, offs)
    assert not st.any()
    one_t, one_off, _ = toks[0].tokenize_packed(utf8, offs)
    assert np.array_equal(t, one_t) and np.array_equal(toff, one_off)
    # many super-chunks: the pipeline's slots are reused, the chunk boundaries fall everywhere
    monkeypatch.setenv("KGPU_MULTI_CHUNK_SENTS", "7")
    for n in (0, 1, 2, 3, 4, 5, 7, 64, 1000):  # ragged n: empty shards (n < G), a last super-chunk shorter than G
        u, o = pack_sentences(sents[:n])
        _check(toks[:G], orc, u, o)
    _check(toks[:G], orc, utf8, offs)  # ~300 super-chunks of 7 G sentences through the eight slots


@pytest.mark.parametrize("G", [1, 2, 3])
def test_compact_sharded_call_expands_to_the_same_records(env, G, monkeypatch):
    """kgpu_tokenize_batch_multi_compact: 8-byte records + firsts + offsets in the caller's order; kgpu_expand_tokens over them is byte for byte the
    24-byte form's result and the oracle's, at G = 1 (the pipeline with one shard), 2 and 3, with many small super-chunks, ragged n and a bad sentence."""
    from kanpyo_amd import _lib, synth
    from kanpyo_amd.device import expand_tokens
    from kanpyo_amd.tokenizer import TOKEN8_DTYPE, pack_sentences, tokenize_packed_multi

    sd, toks, orc = env
    sents = synth.make_corpus(sd, 5000, 13, "cfg2") + synth.make_corpus(sd, 200, 14, "cfg3") + ["", "すもももももももものうち", "テ"]
    np.random.default_rng(40 + G).shuffle(sents)

    def check(ss):
        utf8, offs = pack_sentences(ss)
        t8, first, toff, st = tokenize_packed_multi(toks[:G], utf8, offs, compact=True)
        exp = orc.tokenize_batch(utf8, offs, 8)
        assert t8.dtype == TOKEN8_DTYPE and np.array_equal(toff, exp.offsets) and not st.any()
        assert np.array_equal(expand_tokens(t8, toff, first), exp.tokens)
        return utf8, offs

    utf8, offs = check(sents)
    out = (np.empty(10, dtype=TOKEN8_DTYPE), np.empty(len(offs), dtype=np.uint64), np.empty(len(offs), dtype=np.uint8))
    with pytest.raises(_lib.KgpuError) as e:
        tokenize_packed_multi(toks[:G], utf8, offs, out=out, compact=True)
    assert e.value.code == _lib.KGPU_ERR_CAPACITY
    monkeypatch.setenv("KGPU_MULTI_CHUNK_SENTS", "7")
    for n in (0, 1, 2, 3, 5, 64, 1000):
        check(sents[:n])
    raw = [s.encode("utf-8") for s in sents[:500]]
    raw[17] = b"\xff\xfe broken"
    o = np.concatenate([[0], np.cumsum([len(s) for s in raw])]).astype(np.uint64)
    u = np.frombuffer(b"".join(raw), dtype=np.uint8)
    t8, first, toff, st = tokenize_packed_multi(toks[:G], u, o, compact=True)
    one_t, one_off, one_st = toks[0].tokenize_packed(u, o)
    assert st[17] == 1 and np.array_equal(st, one_st) and np.array_equal(toff, one_off) and np.array_equal(expand_tokens(t8, toff, first), one_t)


def test_same_handle_several_times_and_status_bytes(env):
    from kanpyo_amd.tokenizer import tokenize_packed_multi

    sd, toks, orc = env
    from kanpyo_amd import synth

    sents = [s.encode("utf-8") for s in synth.make_corpus(sd, 2000, 9, "cfg2")]
    sents[17] = b"\xff\xfe broken"       # invalid UTF-8: flagged, zero tokens, the others unaffected
    sents[1501] = b"\xe3\x81"
    offs = np.concatenate([[0], np.cumsum([len(s) for s in sents])]).astype(np.uint64)
    utf8 = np.frombuffer(b"".join(sents), dtype=np.uint8)
    t, toff, st = tokenize_packed_multi([toks[0], toks[0], toks[1], toks[0]], utf8, offs)
    assert st[17] == 1 and st[1501] == 1 and st.sum() == 2
    assert toff[18] == toff[17] and toff[1502] == toff[1501]
    one_t, one_off, one_st = toks[0].tokenize_packed(utf8, offs)
    assert np.array_equal(t, one_t) and np.array_equal(toff, one_off) and np.array_equal(st, one_st)


def test_capacity_error_reports_the_size_needed(env):
    from kanpyo_amd import _lib, synth
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences, tokenize_packed_multi

    sd, toks, orc = env
    utf8, offs = pack_sentences(synth.make_corpus(sd, 3000, 5, "cfg2"))
    exp = orc.tokenize_batch(utf8, offs, 8)
    out = (np.empty(100, dtype=TOKEN_DTYPE), np.empty(len(offs), dtype=np.uint64), np.empty(len(offs), dtype=np.uint8))
    with pytest.raises(_lib.KgpuError) as e:
        tokenize_packed_multi(toks[:2], utf8, offs, out=out)
    assert e.value.code == _lib.KGPU_ERR_CAPACITY and "need %d" % len(exp.tokens) in str(e.value)
    t, toff, _ = tokenize_packed_multi(toks[:2], utf8, offs)  # the wrapper retries with the reported size
    assert np.array_equal(t, exp.tokens)


def test_device_resident_shards_gathered_on_the_root(env):
    """kgpu_multi_*: shard g resident on entry g's device, the compaction kernels store the 8-byte records into the ROOT's memory (here every
    entry is device 0: the stores are local, the code path is the same), kgpu_expand_tokens restores the records; reassembled == oracle."""
    import torch

    from kanpyo_amd import _lib, synth
    from kanpyo_amd.dist import reassemble
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences

    sd, toks, orc = env
    L = _lib.lib()
    G = 3
    sents = synth.make_corpus(sd, 5000, 21, "cfg2") + synth.make_corpus(sd, 100, 22, "cfg3")
    utf8, offs = pack_sentences(sents)
    exp = orc.tokenize_batch(utf8, offs, 8)
    dev = torch.device("cuda", 0)
    handles = (C.c_void_p * G)(*[t.handle for t in toks])
    mh = C.c_void_p()
    _lib.check(L.kgpu_multi_create(handles, G, 2, C.byref(mh)))
    try:
        shards = []
        for g in range(G):
            u, o = pack_sentences(sents[g::G])
            n, total = len(o) - 1, int(o[-1])
            cap = total + n + 1
            shards.append(dict(u=torch.from_numpy(u.copy()).to(dev), o=torch.from_numpy(o.astype(np.int64)).to(dev), n=n, total=total, cap=cap,
                               t8=torch.empty((cap, 2), dtype=torch.int32, device=dev), first=torch.empty(2 * n, dtype=torch.int32, device=dev),
                               toff=torch.empty(n + 1, dtype=torch.int64, device=dev), st=torch.empty(n + 16, dtype=torch.uint8, device=dev)))
        arr = lambda key: (C.c_void_p * G)(*[s[key].data_ptr() for s in shards])
        u64 = lambda key: (C.c_uint64 * G)(*[s[key] for s in shards])
        for slot in (0, 1, 0):
            _lib.check(L.kgpu_multi_tokenize_device(mh, slot, arr("u"), arr("o"), u64("n"), u64("total"), arr("t8"), u64("cap"), arr("first"), arr("toff"), arr("st")))
            got = (C.c_uint64 * G)()
            _lib.check(L.kgpu_multi_sync(mh, slot, got))
            toks24, counts = [], []
            for g, s in enumerate(shards):
                toff = s["toff"].cpu().numpy().astype(np.uint64)
                assert int(got[g]) == int(toff[-1])
                t8 = np.ascontiguousarray(s["t8"][: int(got[g])].cpu().numpy())
                first = np.ascontiguousarray(s["first"].cpu().numpy().astype(np.uint32))
                out = np.empty(int(got[g]), dtype=TOKEN_DTYPE)
                L.kgpu_expand_tokens(t8.ctypes.data, toff.ctypes.data, first.ctypes.data, s["n"], out.ctypes.data)
                assert not s["st"][: s["n"]].any()
                toks24.append(out.view(np.int32).reshape(-1, 6))
                counts.append(np.diff(toff).astype(np.int64))
            g_tok, g_off = reassemble(np.concatenate(toks24), np.concatenate(counts), len(sents), G)
            assert np.array_equal(g_off.astype(np.uint64), exp.offsets)
            assert np.array_equal(g_tok.reshape(-1), exp.tokens.view(np.int32).reshape(-1))
    finally:
        L.kgpu_multi_destroy(mh)


def test_bench_single_process_path_on_one_gpu():
    """bench.py --gpus 2 --single-process --devices 0,0: cfg 4's workload through kgpu_multi_* in one process (no torch.distributed); the line has the
    multi-rank path's fields, and the run itself asserts that a gathered + expanded + reassembled step equals the single-device token stream."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0", "--steps", "2", "--warmup", "1",
                        "--prewarm-seconds", "0.05", "--corpora", "1", "--queue", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "config", "sentences_total", "gather", "per_rank_sentences_per_s", "full_report"):
        assert k in line, k
    assert line["n_gpus"] == 2 and line["sentences_total"] == 200_000 and line["gather"]["reassembled_step_equals_one_gpu"] is True
    assert line["value"] > 1e6


def test_one_handle_per_device():
    """Both multi-device forms with one dictionary handle per PHYSICAL device (advisor, round 4: everything above passes device 0 several times, so
    hipDeviceEnablePeerAccess, the compaction kernels' stores into another device's memory and one pipeline thread per device have never executed).
    Skipped on a one-GPU box -- which is every box this repository has been run on so far: the multi-device API is experimental until this has passed."""
    import torch

    from kanpyo_amd import Tokenizer, _lib, synth
    from kanpyo_amd.dist import reassemble
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences
    from oracle import oracle

    L = _lib.lib()
    ndev = L.kgpu_device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} HIP device(s): the peer-access gather and the per-device pipelines need two")
    G = min(ndev, 4)
    oracle.build()
    sd = synth.build_dict(20000, seed=11)
    orc = oracle.OracleTokenizer.from_dict(sd.dict)
    toks = [Tokenizer(sd.dict, device=g) for g in range(G)]
    before = torch.cuda.current_device()
    sents = synth.make_corpus(sd, 9000, 31, "cfg2") + synth.make_corpus(sd, 200, 32, "cfg3") + ["", "テ"]
    utf8, offs = pack_sentences(sents)
    exp = orc.tokenize_batch(utf8, offs, 8)
    _check(toks, orc, utf8, offs)                                   # kgpu_tokenize_batch_multi: host buffers, one pipeline thread per device
    _check(toks[:1] + toks[:1] + toks[1:2], orc, utf8, offs)        # a device twice and another once
    assert torch.cuda.current_device() == before                    # the caller's current device is put back (advisor, round 4)
    handles = (C.c_void_p * G)(*[t.handle for t in toks])
    mh = C.c_void_p()
    _lib.check(L.kgpu_multi_create(handles, G, 2, C.byref(mh)))     # peer access from every device to the root's memory
    try:
        root = torch.device("cuda", 0)
        shards = []
        for g in range(G):
            dev = torch.device("cuda", g)
            u, o = pack_sentences(sents[g::G])
            n, total = len(o) - 1, int(o[-1])
            cap = total + n + 1
            shards.append(dict(u=torch.from_numpy(u.copy()).to(dev), o=torch.from_numpy(o.astype(np.int64)).to(dev), n=n, total=total, cap=cap,
                               t8=torch.empty((cap, 2), dtype=torch.int32, device=root), first=torch.empty(2 * n, dtype=torch.int32, device=root),
                               toff=torch.empty(n + 1, dtype=torch.int64, device=root), st=torch.empty(n + 16, dtype=torch.uint8, device=root)))
        torch.cuda.synchronize()
        arr = lambda key: (C.c_void_p * G)(*[s[key].data_ptr() for s in shards])
        u64 = lambda key: (C.c_uint64 * G)(*[s[key] for s in shards])
        for slot in (0, 1, 0):
            _lib.check(L.kgpu_multi_tokenize_device(mh, slot, arr("u"), arr("o"), u64("n"), u64("total"), arr("t8"), u64("cap"), arr("first"), arr("toff"), arr("st")))
            got = (C.c_uint64 * G)()
            _lib.check(L.kgpu_multi_sync(mh, slot, got))
            toks24, counts = [], []
            for g, s in enumerate(shards):
                toff = s["toff"].cpu().numpy().astype(np.uint64)
                assert int(got[g]) == int(toff[-1])
                t8 = np.ascontiguousarray(s["t8"][: int(got[g])].cpu().numpy())
                first = np.ascontiguousarray(s["first"].cpu().numpy().astype(np.uint32))
                out = np.empty(int(got[g]), dtype=TOKEN_DTYPE)
                L.kgpu_expand_tokens(t8.ctypes.data, toff.ctypes.data, first.ctypes.data, s["n"], out.ctypes.data)
                assert not s["st"][: s["n"]].any()
                toks24.append(out.view(np.int32).reshape(-1, 6))
                counts.append(np.diff(toff).astype(np.int64))
            g_tok, g_off = reassemble(np.concatenate(toks24), np.concatenate(counts), len(sents), G)
            assert np.array_equal(g_off.astype(np.uint64), exp.offsets)
            assert np.array_equal(g_tok.reshape(-1), exp.tokens.view(np.int32).reshape(-1))
    finally:
        L.kgpu_multi_destroy(mh)
        for t in toks:
            t.close()

"""Concurrent callers of kgpu_tokenize_batch: the reference's tokenize() takes &self and is Send + Sync (src/tokenizer.rs:16), and its call
shape is ONE sentence per call (src/bin/kanpyo.rs:106-126) -- a server calls it from many threads.  Small calls that arrive together share a
launch (the combiner, kgpu_api.cpp); every caller must still get exactly its own sentences' records, bit-exact against the oracle."""
import threading

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
    return sd, Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)


def _run_threads(tok, orc, work, nthreads):
    """work[t] = list of (utf8, offs) calls of thread t; returns the list of failures."""
    from kanpyo_amd.tokenizer import TOKEN_DTYPE

    errors = []
    start = threading.Barrier(nthreads)

    def body(t):
        try:
            out = (np.empty(1 << 16, dtype=TOKEN_DTYPE), np.empty(4200, dtype=np.uint64), np.empty(4200, dtype=np.uint8))
            start.wait()
            for k, (u, o) in enumerate(work[t]):
                got_t, got_o, st = tok.tokenize_packed(u, o, out=out)
                exp = orc.tokenize_batch(u, o, 1)
                if st.any() or not np.array_equal(got_o, exp.offsets) or not np.array_equal(got_t, exp.tokens):
                    errors.append((t, k, len(o) - 1))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=body, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return errors


def test_32_threads_mixed_call_sizes_bit_exact(env):
    """Python threads (the GIL serialises what surrounds each call, so few calls meet inside the library): every caller gets its own records."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = env
    sents = synth.make_corpus(sd, 4000, 31, "cfg2") + synth.make_corpus(sd, 200, 32, "cfg3") + ["", "テ", "すもももももももものうち"] * 5
    rng = np.random.default_rng(77)
    nthreads = 32
    work = []
    for t in range(nthreads):
        calls = []
        for _ in range(40):
            n = int(rng.choice([1, 1, 1, 1, 2, 3, 5, 17, 64, 128, 129, 700]))  # mostly the reference's n = 1; some beyond the single-launch path
            idx = rng.integers(0, len(sents), size=n)
            calls.append(pack_sentences([sents[i] for i in idx]))
        work.append(calls)
    tok.routing(reset=True)
    errors = _run_threads(tok, orc, work, nthreads)
    assert not errors, errors[:5]
    assert tok.routing()["small_calls"] > 0


def test_native_threads_share_launches_and_stay_bit_exact(env):
    """32 and 64 NATIVE host threads (kgpu_debug_concurrent_callers: no GIL between the calls) x mixed n: calls do meet inside the entry point, share
    launches (kgpu_routing.combined_calls), and every call's records equal the oracle's for exactly its sentences."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import concurrent_callers, pack_sentences

    sd, tok, orc = env
    sents = synth.make_corpus(sd, 3000, 61, "cfg2") + synth.make_corpus(sd, 150, 62, "cfg3") + ["", "テ", "すもももももももものうち"] * 5
    np.random.default_rng(3).shuffle(sents)
    utf8, offs = pack_sentences(sents)
    exp = orc.tokenize_batch(utf8, offs, 8)
    # (128 x n = 1: a batch's two dozen followers are released by one wake-up and reach the combiner's lock together)
    for threads, pattern, calls in ((32, (1, 1, 1, 2, 5, 1, 17, 1, 64, 1, 130, 1), 60), (64, (1,), 100), (8, (128, 1), 40), (128, (1,), 60), (96, (1, 2, 1, 3), 40)):
        tok.routing(reset=True)
        r = concurrent_callers(tok, utf8, offs, threads, calls, pattern, expect=(exp.tokens, exp.offsets))
        assert r["mismatching_calls"] == 0 and r["calls"] == threads * calls, r
        rt = tok.routing()
        assert rt["small_calls"] > 0
        if threads >= 32:
            assert rt["combined_calls"] >= 2 and 1 <= rt["combined_launches"] < rt["combined_calls"], rt


def test_n1_from_64_threads_and_a_long_sentence_among_them(env):
    """A sentence the single-launch path cannot serve (too long for the LDS pool) sends only ITS caller down the general path... after the whole
    combined launch reported the hand-over: every caller of that launch must still come back with its own correct records."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = env
    short = synth.make_corpus(sd, 500, 41, "cfg2")
    long_ = synth.make_corpus(sd, 8, 42, "cfg5")  # 2048-character documents
    rng = np.random.default_rng(5)
    nthreads = 64
    work = []
    for t in range(nthreads):
        calls = [pack_sentences([short[int(rng.integers(0, len(short)))]]) for _ in range(30)]
        if t % 16 == 3:
            calls.insert(7, pack_sentences([long_[t % len(long_)][:900]]))  # ~2.7 KB: beyond the pool kernel's routing limit, within the small call's 16 KB
        work.append(calls)
    errors = _run_threads(tok, orc, work, nthreads)
    assert not errors, errors[:5]


def test_capacity_error_is_per_caller(env):
    """Two callers in one combined launch, one with a token buffer that is too small: only that one gets KGPU_ERR_CAPACITY (with the size it needs)."""
    from kanpyo_amd import _lib, synth
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences

    sd, tok, orc = env
    sents = synth.make_corpus(sd, 64, 51, "cfg2")
    results = {}
    start = threading.Barrier(8)

    def body(t):
        u, o = pack_sentences([sents[t]])
        cap = 1 if t == 0 else 256
        out = (np.empty(cap, dtype=TOKEN_DTYPE), np.empty(2, dtype=np.uint64), np.empty(1, dtype=np.uint8))
        start.wait()
        for _ in range(50):
            try:
                got = tok.tokenize_packed(u, o, out=out)
                results.setdefault(t, []).append(("ok", len(got[0])))
            except _lib.KgpuError as e:
                results.setdefault(t, []).append(("err", e.code, str(e)))

    ths = [threading.Thread(target=body, args=(t,)) for t in range(8)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    exp = {t: len(orc.tokenize_batch(*pack_sentences([sents[t]]), 1).tokens) for t in range(8)}
    assert all(r[0] == "err" and r[1] == _lib.KGPU_ERR_CAPACITY and f"need {exp[0]}" in r[2] for r in results[0]), results[0][:3]
    for t in range(1, 8):
        assert all(r == ("ok", exp[t]) for r in results[t]), (t, results[t][:3])

"""GPU parity: the HIP path (through the C ABI) must be bit-exact against the CPU
oracle on the same inputs.  Integer/byte work => exact equality, no tolerance.
Run on the GPU box with `pytest -m gpu`."""
import numpy as np
import pytest

from conftest import fixture_dict_parts, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    from kanpyo_amd import _lib

    assert _lib.lib().kgpu_device_count() > 0, "no HIP device: the gpu tests need an MI355X"
    from oracle import oracle

    oracle.build()
    return _lib, oracle


@pytest.fixture(scope="module")
def small(libs):
    """20k-record synthetic dictionary: GPU tokenizer + oracle over the same blobs."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    sd = synth.build_dict(20000, seed=11)
    return sd, Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)


@pytest.fixture(scope="module")
def full(libs):
    """The 392k-record IPADIC-shaped dictionary of BASELINE.json's configs."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    sd = synth.build_dict()
    return sd, Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)


def assert_same(tok, orc, sentences, nthreads=8):
    from kanpyo_amd.tokenizer import pack_sentences

    utf8, offs = pack_sentences(sentences)
    got_t, got_off, status = tok.tokenize_packed(utf8, offs)
    exp = orc.tokenize_batch(utf8, offs, nthreads)
    assert np.array_equal(status, np.zeros(len(sentences), dtype=np.uint8))
    assert np.array_equal(got_off, exp.offsets), "per-sentence token counts differ"
    if not np.array_equal(got_t, exp.tokens):
        bad = np.nonzero(got_t != exp.tokens)[0][0]
        s = int(np.searchsorted(exp.offsets, bad, side="right") - 1)
        raise AssertionError(f"token {bad} (sentence {s}: {sentences[s]!r}) differs: gpu {got_t[bad]} oracle {exp.tokens[bad]}")
    return exp


def test_fixture_vectors(libs):
    """cfg 1 plumbing: the reference's own fixture dictionary + App. C vectors."""
    from kanpyo_amd import Dict, Tokenizer

    tok = Tokenizer(Dict.from_parts(**fixture_dict_parts()))
    for case in load_golden("fixture_tokens.json")["cases"]:
        got = tok.tokenize(case["input"])
        assert [[t.id, int(t.class_), t.position, t.start, t.end, t.surface] for t in got] == case["tokens"], case["input"]
    # the reference's own assertions (src/tests.rs:110-171)
    toks = tok.tokenize("テスト")
    assert toks and any(t.class_ != 0 for t in toks)
    assert all(t.start <= t.end <= 3 for t in toks if t.class_ != 0)
    assert tok.tokenize("") and tok.tokenize("あいうえお")


def test_cfg1_literal_sentence(full):
    sd, tok, orc = full
    s = "すもももももももものうち"
    exp = assert_same(tok, orc, [s])
    toks = tok.tokenize(s)
    assert toks[-1].surface == "EOS" and toks[-1].position == 36 and toks[-1].start == 12 and toks[-1].end == 15
    assert "".join(t.surface for t in toks[:-1]) == s  # hiragana: every char is a word in the synthetic lexicon


def test_small_dict_corpora(small):
    from kanpyo_amd import synth

    sd, tok, orc = small
    assert_same(tok, orc, synth.make_corpus(sd, 3000, 1, "cfg2"))
    assert_same(tok, orc, synth.make_corpus(sd, 600, 2, "cfg3"))
    assert_same(tok, orc, synth.make_corpus(sd, 6, 5, "cfg5"))


def test_cfg2_full_dict(full):
    from kanpyo_amd import synth

    sd, tok, orc = full
    exp = assert_same(tok, orc, synth.make_corpus(sd, 20000, 1, "cfg2"))
    assert exp.counters["K"] > 20000


def test_cfg3_mixed_lengths_unknown_heavy(full):
    from kanpyo_amd import synth

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 4000, 2, "cfg3")
    assert any(any(ord(c) >= 0x10000 for c in s) for s in sents)  # non-BMP -> table[0]
    assert_same(tok, orc, sents)


def test_cfg5_long_documents(full):
    from kanpyo_amd import synth

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 24, 5, "cfg5")
    assert all(len(s) == 2048 for s in sents)
    assert_same(tok, orc, sents)


def test_cfg3_and_cfg5_at_scale(full):
    """The routing between the pool kernel and the windowed kernel under a realistic mix, several batches
    deep (the reservation estimate and the optional-launch heuristics adapt from batch to batch)."""
    from kanpyo_amd import synth

    sd, tok, orc = full
    mixed = synth.make_corpus(sd, 20000, 7, "cfg3")
    for lo in range(0, len(mixed), 4096):
        assert_same(tok, orc, mixed[lo : lo + 4096], nthreads=16)
    assert_same(tok, orc, synth.make_corpus(sd, 300, 8, "cfg5") + synth.make_corpus(sd, 3000, 9, "cfg2"), nthreads=16)


def test_one_very_long_sentence(full):
    """100 000 characters in one sentence (over a million lattice nodes: 32-bit node indices, global
    counters instead of the LDS cursors, many backtrace windows) next to ordinary ones."""
    from kanpyo_amd import synth

    sd, tok, orc = full
    parts = synth.make_corpus(sd, 2600, 13, "cfg2")
    long_one = "".join(parts)[:100000]
    assert len(long_one) == 100000
    assert_same(tok, orc, ["短い文", long_one, "", "もう一つ"], nthreads=2)


def test_edge_cases(full):
    sd, tok, orc = full
    kata = "ア" * 1500  # one groupable run beyond MAXIMUM_UNKNOWN_WORD_LENGTH (lattice.rs:55,80)
    cases = [
        "", "あ", "ア", "a", "0", " ", "　", "𠮷", "𠮷野家", "a\x00b", "\x00", "アアア", kata, "1" * 1024, "1" * 1025,
        "x" * 1023 + "あ", "漢" * 300, "あ" * 700, "ＡＢＣ１２３", "Ωμέγα", "привет", "東京都に住む。", "、。「」",
        "あ𠮷" * 40, "\t\n", "é" * 50,
    ]
    assert_same(tok, orc, cases)
    # ragged batch: empties interleaved, single sentence, repeated sentence
    assert_same(tok, orc, ["", "", "すもも", "", "もも", ""])
    assert_same(tok, orc, ["すもももももももものうち"] * 257)


def test_byte_length_and_alignment_boundaries(full):
    """Every byte length from 0 to 300 at every alignment of its first byte in the batch (the pool kernel takes a sentence of up to ~250 bytes as one aligned
    dword per lane and byte-aligns it against the next lane's; longer ones byte by byte), sentences of more than 64 tokens (the successor's start position
    comes from the next lane, the 64th token's from the next round), of 255 / 256 / 257 characters (the start-position search has eight fixed steps up to 255),
    and lattices of 1 .. ~130 tiles (stage B's windows of 64 descriptors, a last group of 1 .. 8 tiles) -- all against the oracle."""
    sd, tok, orc = full
    from kanpyo_amd import synth

    base = [t for t in synth.make_corpus(sd, 400, 5, "cfg3") if len(t) >= 110][:64]   # dictionary words and unknown runs, 110+ characters each
    assert len(base) == 64
    sents = []
    for L in range(0, 301):
        body = base[L % len(base)]
        cut = body.encode("utf-8")[: (L // 3) * 3].decode("utf-8", errors="ignore")   # whole characters of the corpus' text ...
        cut += "a1b"[: L - len(cut.encode("utf-8"))] if len(cut.encode("utf-8")) <= L else ""
        while len(cut.encode("utf-8")) < L:
            cut += "z"                                                                   # ... topped up to the byte length with ASCII
        assert len(cut.encode("utf-8")) == L
        sents.append(cut)
    for pad in ("", "x", "xy", "xyz"):                 # shifts every later sentence's first byte through the four alignments
        assert_same(tok, orc, [pad] + sents)
    many_tokens = ["あ1" * n for n in (31, 32, 33, 63, 64, 65, 100)] + ["a あ" * 50]
    chars = ["あ" * n for n in (254, 255, 256, 257)] + ["1a" * 128, "1a" * 127 + "1"]
    assert_same(tok, orc, many_tokens + chars)
    assert_same(tok, orc, [base[i][:n] for i, n in enumerate(range(1, 64))])   # 1 .. 63 characters: tile counts around the group and window sizes


def test_invalid_utf8_is_flagged_not_tokenized(full):
    from kanpyo_amd import _lib
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    sents = [b"ok", b"\xff", "あ".encode(), b"\xe3\x81", b"\x80abc", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", "終".encode()]
    utf8, offs = pack_sentences(sents)
    t, toff, status = tok.tokenize_packed(utf8, offs)
    assert status.tolist() == [0, 1, 0, 1, 1, 1, 1, 1, 0]
    for i, s in enumerate(sents):
        if status[i]:
            assert toff[i + 1] == toff[i]
            with pytest.raises(UnicodeDecodeError):
                orc.tokenize(s)
        else:
            exp, _ = orc.tokenize(s)
            assert np.array_equal(t[int(toff[i]) : int(toff[i + 1])], exp)
    assert _lib.KGPU_SENT_INVALID_UTF8 == 1


def test_unreachable_eos_and_dead_ends(libs):
    """Viterbi quirks on hand-made dictionaries."""
    from kanpyo_amd import Dict, Tokenizer

    _, oracle = libs
    p = fixture_dict_parts()
    # negative connection cost lets an UNREACHABLE predecessor win (INF + cost + matrix < INF)
    p["conn_data"] = [0, 100, 200, 100, -30000, 100, 200, 100, -30000]
    p["morphs"] = [[0, 0, 1000], [1, 1, -20000], [2, 2, 1100]]
    d = Dict.from_parts(**p)
    tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
    assert_same(tok, orc, ["テ", "テあ", "テ辞書", "テ辞書形態素", "テスト辞書", "ト辞書あ", "辞書テ", "形態素テ形態素", "テテ辞書辞書"], nthreads=1)


def test_duplicate_heavy_and_deep_prefix_dictionary(libs):
    from kanpyo_amd import Dict, Tokenizer

    _, oracle = libs
    rng = np.random.default_rng(3)
    kws = []
    for k in range(1, 41):  # every prefix of a 40-char string is a word, each with many records
        kws += ["あ" * k] * int(rng.integers(1, 30))
    kws += ["い"] * 300
    kws.sort(key=lambda s: s.encode())
    morphs = np.stack([rng.integers(0, 8, len(kws)), rng.integers(0, 8, len(kws)), rng.integers(-500, 9000, len(kws))], axis=1)
    p = fixture_dict_parts()
    d = Dict.from_parts(kws, morphs, 8, 8, rng.integers(-800, 800, 64), p["char_class"], p["char_category"],
                        p["invoke_list"], p["group_list"], {1: (1, 2), 2: (1, 2)}, [[0, 0, 3000], [1, 1, 3500]])
    tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
    assert_same(tok, orc, ["あ" * n for n in (1, 2, 39, 40, 41, 100)] + ["い" * 70, "あいあいあ" * 30, "いあ" * 64], nthreads=1)


def test_idempotent_and_order_independent(full):
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 5000, 9, "cfg2")
    a = tok.tokenize_packed(*pack_sentences(sents))
    b = tok.tokenize_packed(*pack_sentences(sents))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    perm = np.random.default_rng(0).permutation(len(sents))
    c = tok.tokenize_packed(*pack_sentences([sents[i] for i in perm]))
    for new_i, old_i in enumerate(perm[:500]):
        assert np.array_equal(c[0][int(c[1][new_i]) : int(c[1][new_i + 1])], a[0][int(a[1][old_i]) : int(a[1][old_i + 1])])


def test_full_size_properties_cfg2(full):
    """BASELINE configs[1] at full size (100k sentences, batches of 4096): size-
    independent properties on all of it, oracle equality on a bounded sample."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 100_000, 1, "cfg2")
    checked = 0
    for lo in range(0, len(sents), 4096):
        chunk = sents[lo : lo + 4096]
        utf8, offs = pack_sentences(chunk)
        t, toff, status = tok.tokenize_packed(utf8, offs)
        assert not status.any()
        nb = (offs[1:] - offs[:-1]).astype(np.int64)
        cnt = (toff[1:] - toff[:-1]).astype(np.int64)
        assert (cnt >= 1).all()  # every category has an unk entry => EOS reachable
        last = t[toff[1:].astype(np.int64) - 1]
        assert (last["cls"] == 0).all() and (last["id"] == 0).all() and (last["byte_len"] == 0).all()
        assert np.array_equal(last["position"].astype(np.int64), nb)  # EOS.position == B
        assert np.array_equal(last["end"], last["start"] + 3)
        # tokens tile the sentence: position[k+1] == position[k] + byte_len[k], first at 0
        sent_of = np.repeat(np.arange(len(chunk)), cnt)
        first = np.zeros(len(t), dtype=bool); first[toff[:-1].astype(np.int64)] = True
        assert (t["position"][first] == 0).all() and (t["start"][first] == 0).all()
        nxt = ~first
        assert np.array_equal(t["position"][nxt], (t["position"] + t["byte_len"])[np.nonzero(nxt)[0] - 1])
        words = t["cls"] != 0
        assert np.array_equal(t["start"][nxt], t["end"][np.nonzero(nxt)[0] - 1])
        assert (t["id"][words] >= 1).all() and sent_of.size == len(t)
        if lo % (4096 * 6) == 0:  # oracle equality on every 6th batch
            exp = orc.tokenize_batch(utf8, offs, 8)
            assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens)
            checked += 1
    assert checked >= 4


def test_full_size_dense_dictionary(libs):
    """The dense-lattice variant of the 392k-record dictionary (natural density, SURVEY 8a a15: N ~ 8-10 x C; more than eight predecessors at 61 % of the
    positions -- target groups of several tiles, chunks that combine, three wavefronts per pool by the runtime's rule): cfg 2-shaped text at full size
    (100k sentences, batches of 4096), the tiling properties on all of it, oracle equality on every 6th batch."""
    from kanpyo_amd import Tokenizer, synth
    from kanpyo_amd.tokenizer import pack_sentences

    _, oracle = libs
    sd = synth.build_dict(dense=True)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    sents = synth.make_corpus(sd, 100_000, 1, "cfg2")
    checked = 0
    for lo in range(0, len(sents), 4096):
        utf8, offs = pack_sentences(sents[lo : lo + 4096])
        t, toff, status = tok.tokenize_packed(utf8, offs)
        assert not status.any()
        cnt = (toff[1:] - toff[:-1]).astype(np.int64)
        assert (cnt >= 1).all()
        last = t[toff[1:].astype(np.int64) - 1]
        assert (last["cls"] == 0).all() and np.array_equal(last["position"].astype(np.int64), (offs[1:] - offs[:-1]).astype(np.int64))
        first = np.zeros(len(t), dtype=bool); first[toff[:-1].astype(np.int64)] = True
        nxt = ~first
        assert (t["position"][first] == 0).all()
        assert np.array_equal(t["position"][nxt], (t["position"] + t["byte_len"])[np.nonzero(nxt)[0] - 1])
        if lo % (4096 * 6) == 0:
            exp = orc.tokenize_batch(utf8, offs, 8)
            assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens)
            checked += 1
    assert checked >= 4
    tok.close()


def _device_run(tok, sentences, mode):
    """Drive the device-resident C ABI (kgpu_ctx_*) with torch-owned HBM buffers."""
    import torch

    from kanpyo_amd.device import DeviceContext
    from kanpyo_amd.tokenizer import pack_sentences

    utf8, offs = pack_sentences(sentences)
    dev = torch.device("cuda", 0)
    d_utf8 = torch.from_numpy(utf8.copy()).to(dev) if utf8.size else torch.zeros(1, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    n, cap = len(sentences), int(offs[-1]) + len(sentences)
    d_tok = torch.empty((cap, 6), dtype=torch.int32, device=dev)
    d_toff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    ctx = DeviceContext(tok)
    ctx.set_profiling(mode)
    ctx.tokenize(d_utf8.data_ptr(), d_off.data_ptr(), n, int(offs[-1]), d_tok.data_ptr(), cap, d_toff.data_ptr(), d_st.data_ptr())
    nt = ctx.sync()
    return ctx, d_tok[:nt].cpu().numpy(), d_toff.cpu().numpy(), utf8, offs


def test_device_api_and_work_counters(full):
    """The device-side work counters (B, C, T, N, E, K) equal the oracle's."""
    from kanpyo_amd import synth
    from kanpyo_amd.device import PROFILE_EVENTS, PROFILE_WORK

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 4096, 1, "cfg2") + synth.make_corpus(sd, 200, 2, "cfg3") + synth.make_corpus(sd, 3, 5, "cfg5")
    ctx, t, toff, utf8, offs = _device_run(tok, sents, PROFILE_WORK | PROFILE_EVENTS)
    exp = orc.tokenize_batch(utf8, offs, 8)
    assert np.array_equal(toff.astype(np.uint64), exp.offsets)
    assert np.array_equal(t.reshape(-1), exp.tokens.view(np.int32).reshape(-1))
    assert ctx.work() == exp.counters
    p = ctx.profile()
    assert p["launches"] == 1 and p["tokenize_ms"] > 0  # (a batch that finds the chain's tail left out gets the tail alone afterwards: the chain is timed once)


@pytest.mark.parametrize("pool,window_kib", [("0", "0"), ("0", "12"), ("0", "8"), ("0", "64"), ("80:10", "32"), ("80:8:20", "12"), ("80:8,160:4", "24"),
                                             ("16:4:8", "16"), ("8:2,160:2", "0"), ("160:16:64", "10"), ("40:16,24:1", "9"), ("24:1", "48")])
def test_every_launch_chain_is_bit_exact(libs, pool, window_kib, monkeypatch):
    """Force sentences through every launch chain.  KGPU_POOL = KiB of LDS per workgroup : independent
    wavefronts sharing it [: pages of 64 a sentence may take] ('0' = no pool kernel), under pressure: more wavefronts than the pool can serve
    at once, pools too small for the long sentences, reservations that prove too small (redo).  KGPU_WINDOW = KiB of LDS of the windowed
    kernel behind the pools ('0' = off: the general, HBM-scratch kernel serves everything the pools route away; small = windows are halved
    again and again and the kernel hands on what does not fit even four positions long).  Whatever is left ends in the general kernel."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    monkeypatch.setenv("KGPU_POOL", pool)
    monkeypatch.setenv("KGPU_WINDOW", window_kib)
    sd = synth.build_dict(20000, seed=11)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    sents = synth.make_corpus(sd, 3000, 3, "cfg2") + synth.make_corpus(sd, 300, 4, "cfg3") + synth.make_corpus(sd, 2, 6, "cfg5") + ["", "あ", "ア" * 1500]
    for _ in range(3):  # the reservation estimate and the optional-launch heuristics adapt between calls
        assert_same(tok, orc, sents)
    assert_same(tok, orc, ["", "", "", "あ", "", "すもも", ""])


def test_built_and_reloaded_dictionary(libs, tmp_path):
    """8(f): a dictionary built from MeCab-format sources, saved as a Kanpyo .dict, reloaded and
    uploaded tokenises bit-exactly like the oracle over the same blobs."""
    from test_builder_cpu import CHAR_DEF, LEX_A, LEX_B, UNK_DEF, _matrix

    from kanpyo_amd import Tokenizer, builder
    from kanpyo_amd.dictfile import format_tokens, load_dict, save_dict

    _, oracle = libs
    df = builder.build(builder.parse_csv(LEX_A) + builder.parse_csv(LEX_B), _matrix(), CHAR_DEF, builder.parse_unk_def(UNK_DEF))
    p = tmp_path / "mini.dict"
    save_dict(df, str(p))
    back = load_dict(str(p))
    tok, orc = Tokenizer(back.dict), oracle.OracleTokenizer.from_dict(df.dict)
    sents = ["東京都に住む", "東京に住む東京都", "トウキョウ 123 に", "", "a,b都", "住", "にににに"]
    assert_same(tok, orc, sents, nthreads=1)
    # 8(f) rank 3: the CLI's lines (src/bin/kanpyo.rs:174-197) from the GPU's tokens == the same lines from the oracle's tokens
    from kanpyo_amd.token import Token, TokenClass

    for text in sents:
        rec, _ = orc.tokenize(text)
        raw = text.encode("utf-8")
        from_oracle = [Token(int(r["id"]), TokenClass(int(r["cls"])), int(r["position"]), int(r["start"]), int(r["end"]),
                             "EOS" if int(r["cls"]) == 0 else raw[int(r["position"]) : int(r["position"]) + int(r["byte_len"])].decode("utf-8"))
                       for r in rec]
        got = tok.tokenize(text)
        assert got == from_oracle, text
        assert format_tokens(got, back) == format_tokens(from_oracle, df), text
    lines = format_tokens(tok.tokenize("東京都に住む"), back).splitlines()
    assert lines[-1] == "EOS\t" and len(lines) >= 2 and all(l.count("\t") == 1 for l in lines)
    assert any(l.split("\t")[1].endswith("トウキョウト") or "トウキョウト" in l.split("\t")[1] for l in lines[:-1])


def test_host_entry_point_chunks_and_capacity(small, monkeypatch):
    """kgpu_tokenize_batch splits large inputs into bounded chunks; tokens stay dense; a too-small
    buffer reports the exact requirement (KGPU_ERR_CAPACITY)."""
    from kanpyo_amd import _lib, synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = small
    sents = synth.make_corpus(sd, 700, 21, "cfg2") + ["", "あ"] + synth.make_corpus(sd, 40, 22, "cfg3")
    utf8, offs = pack_sentences(sents)
    exp = orc.tokenize_batch(utf8, offs, 4)
    monkeypatch.setenv("KGPU_HOST_CHUNK_BYTES", "3000")  # ~25 sentences per chunk
    t, toff, st = tok.tokenize_packed(utf8, offs)
    assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens) and not st.any()
    with pytest.raises(_lib.KgpuError) as e:
        tok.tokenize_packed(utf8, offs, token_capacity=len(exp.tokens) - 1)
    assert e.value.code == _lib.KGPU_ERR_CAPACITY and str(len(exp.tokens)) in str(e.value)
    t2, toff2, _ = tok.tokenize_packed(utf8, offs, token_capacity=len(exp.tokens))
    assert np.array_equal(t2, exp.tokens)


def test_large_host_call_is_pipelined_and_pinned_buffers_work(small):
    """One host call larger than a chunk (three chunks in flight, results delivered in order) with pageable and
    with pinned (kgpu_host_alloc) buffers."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences, pinned_empty

    sd, tok, orc = small
    sents = synth.make_corpus(sd, 40000, 41, "cfg2") + [""] + synth.make_corpus(sd, 500, 42, "cfg3")
    utf8, offs = pack_sentences(sents)
    exp = orc.tokenize_batch(utf8, offs, 8)
    t, toff, st = tok.tokenize_packed(utf8, offs)
    assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens) and not st.any()
    pu = pinned_empty(utf8.shape, np.uint8); pu[:] = utf8
    po = pinned_empty(offs.shape, np.uint64); po[:] = offs
    t2, toff2, st2 = tok.tokenize_packed(pu, po, pinned=True)
    assert np.array_equal(toff2, exp.offsets) and np.array_equal(t2, exp.tokens) and not st2.any()


def test_concurrent_callers_share_one_dictionary(small):
    """Tokenizer::tokenize takes &self: concurrent calls on one dictionary are legal (src/tokenizer.rs:16)."""
    import threading

    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = small
    jobs = [pack_sentences(synth.make_corpus(sd, 400 + 50 * k, 30 + k, "cfg2")) for k in range(6)]
    want = [orc.tokenize_batch(u, o, 2) for u, o in jobs]
    got = [None] * len(jobs)

    def run(k):
        for _ in range(3):
            got[k] = tok.tokenize_packed(*jobs[k])

    th = [threading.Thread(target=run, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(len(jobs)):
        assert np.array_equal(got[k][1], want[k].offsets) and np.array_equal(got[k][0], want[k].tokens)


def test_lattice_dump_and_graphviz(libs, full):
    """kgpu_lattice_dump reads Lattice{nodes, edges} and the Viterbi state of one sentence back from
    the device; it must equal the naive Python restatement's lattice node for node (insertion order, morphs in the
    dictionary's own context ids, dp, pre, edges ascending), and render to the same DOT (reference src/graphviz.rs)."""
    from kanpyo_amd import Dict, Tokenizer, synth
    from kanpyo_amd.lattice import dump_lattice, graphviz
    from oracle import pyref
    from test_lattice_cpu import lattice_from_pyref

    g = load_golden("fixture_graphviz.json")
    d = Dict.from_parts(**fixture_dict_parts())
    pd = pyref.PyDict(d.index_dict, d.connection_dict, d.morph_dict, d.unk_dict, d.char_category, d.invoke_list, d.group_list)
    tok = Tokenizer(d)
    conn = lambda r, l: pd.conn[pd.row * l + r]  # noqa: E731
    known, unk = (lambda i: g["features"]["known"].get(str(i), ["x"])), (lambda i: g["features"]["unknown"].get(str(i), ["y"]))
    for text in [g["input"], "", "テスト", "テ", "テあ", "あいうえお", "辞書あ辞書"]:
        got, exp = dump_lattice(tok, text), lattice_from_pyref(pd, text)
        assert got.edges == exp.edges, text
        assert got.nodes == exp.nodes, text
        for full_state in (False, True):
            assert graphviz(got, conn, known, unk, 48, full_state) == graphviz(exp, conn, known, unk, 48, full_state)
    assert graphviz(dump_lattice(tok, g["input"]), conn, known, unk, 48, False) == g["dot"]
    # the IPADIC-shaped dictionary (frequency-ranked context ids on the device, duplicates, unknown groups, non-BMP)
    sd, tok2, _ = full
    d2 = sd.dict
    pd2 = pyref.PyDict(d2.index_dict, d2.connection_dict, d2.morph_dict, d2.unk_dict, d2.char_category, d2.invoke_list, d2.group_list)
    for text in synth.make_corpus(sd, 6, 77, "cfg2") + synth.make_corpus(sd, 2, 78, "cfg3")[:1] + ["すもももももももものうち", "ア" * 70 + "\U00020000x"]:
        got, exp = dump_lattice(tok2, text), lattice_from_pyref(pd2, text)
        assert got.edges == exp.edges and got.nodes == exp.nodes, text
        assert [n.key() for n in got.viterbi()] == [n.key() for n in exp.viterbi()]


def test_small_calls_single_launch_path(small, monkeypatch):
    """kgpu_tokenize_batch with a handful of sentences (the reference's own call shape is ONE sentence per call,
    src/bin/kanpyo.rs:106-126) takes the single-launch path: pinned input / output, the pool kernel compacts and
    publishes by itself.  Same records as the oracle for every n up to the path's limit, mixed with what the path must
    hand over to the general one (a sentence too long for LDS) and what it must flag (invalid UTF-8); the capacity error
    reports the exact need."""
    from kanpyo_amd import _lib, synth
    from kanpyo_amd.tokenizer import TOKEN_DTYPE, pack_sentences

    sd, tok, orc = small
    corpus = synth.make_corpus(sd, 400, 21, "cfg2")
    for n in (1, 2, 3, 7, 64, 65, 127, 128, 129):
        assert_same(tok, orc, corpus[:n])
    assert_same(tok, orc, [""])
    assert_same(tok, orc, ["", "すもももももももものうち", ""])
    assert_same(tok, orc, corpus[:5] + ["ア" * 3000] + corpus[5:9])          # one sentence for the windowed kernel: whole call falls back
    assert_same(tok, orc, [corpus[0]] * 128)
    utf8, offs = pack_sentences([corpus[0].encode(), b"\xe3\x81", corpus[1].encode()])
    t, toff, status = tok.tokenize_packed(utf8, offs)
    assert status.tolist() == [0, _lib.KGPU_SENT_INVALID_UTF8, 0] and toff[2] == toff[1]
    exp = orc.tokenize_batch(*pack_sentences([corpus[0], corpus[1]]), 1)
    assert np.array_equal(t, exp.tokens)
    with pytest.raises(_lib.KgpuError) as ei:  # capacity: exact need reported, nothing written past the buffer
        tok.tokenize_packed(*pack_sentences(corpus[:3]), out=(np.empty(2, dtype=TOKEN_DTYPE), np.empty(4, dtype=np.uint64), np.empty(3, dtype=np.uint8)))
    assert ei.value.code == _lib.KGPU_ERR_CAPACITY
    monkeypatch.setenv("KGPU_NO_SMALL_CALLS", "1")  # and the general path gives the same for the same calls
    for n in (1, 7, 128):
        assert_same(tok, orc, corpus[:n])


def test_plain_leaves_layout(small, monkeypatch):
    """A dictionary with 2^21 morphs or more keeps plain leaves (base = -id) and the pool kernel parks its matches in
    two words instead of one; KGPU_PLAIN_LEAVES selects that layout for a small dictionary.  Same records either way
    (index.rs:46-51: the duplicate count then comes from the first morph record)."""
    from kanpyo_amd import Tokenizer, synth

    sd, _, orc = small
    monkeypatch.setenv("KGPU_PLAIN_LEAVES", "1")
    tok = Tokenizer(sd.dict)
    monkeypatch.delenv("KGPU_PLAIN_LEAVES")
    assert_same(tok, orc, synth.make_corpus(sd, 3000, 31, "cfg2"))
    assert_same(tok, orc, synth.make_corpus(sd, 300, 32, "cfg3"))
    assert_same(tok, orc, synth.make_corpus(sd, 5, 33, "cfg2"))  # small-call path


def test_compact_records_device_api(full):
    """kgpu_tokenize_device_compact: 8-byte records + the first token's (position, start) per sentence, expanded on the host by
    kgpu_expand_tokens, equal the 24-byte records of kgpu_tokenize_device (and the oracle's) bit for bit."""
    import torch

    from kanpyo_amd import synth
    from kanpyo_amd.device import DeviceContext, expand_tokens
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 3000, 51, "cfg2") + ["", "あ"] + synth.make_corpus(sd, 120, 52, "cfg3") + synth.make_corpus(sd, 2, 53, "cfg5")
    utf8, offs = pack_sentences(sents)
    dev = torch.device("cuda", 0)
    n, cap = len(sents), int(offs[-1]) + len(sents)
    d_utf8, d_off = torch.from_numpy(utf8.copy()).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_t8 = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    d_first = torch.empty((n, 2), dtype=torch.int32, device=dev)
    d_toff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx = DeviceContext(tok)
    ctx.tokenize_compact(d_utf8.data_ptr(), d_off.data_ptr(), n, int(offs[-1]), d_t8.data_ptr(), cap, d_first.data_ptr(), d_toff.data_ptr(), d_st.data_ptr())
    nt = ctx.sync()
    exp = orc.tokenize_batch(utf8, offs, 8)
    toff = d_toff.cpu().numpy().astype(np.uint64)
    assert nt == len(exp.tokens) and np.array_equal(toff, exp.offsets) and not d_st.cpu().numpy().any()
    got = expand_tokens(d_t8[:nt].cpu().numpy(), toff, d_first.cpu().numpy())
    assert np.array_equal(got, exp.tokens)


def test_large_host_call_compact_pipeline_quirks_and_fallback(libs, small, monkeypatch):
    """The large-call pipeline of kgpu_tokenize_batch (8-byte records into mapped host memory, expanded by worker threads):
    chains whose first token does not start at 0 (an unreachable node heads the best chain and is dropped),
    empty results, a token beyond the 8-byte packing (the chunk falls back to 24-byte records), the legacy form of the whole call."""
    from kanpyo_amd import Dict, Tokenizer, synth
    from kanpyo_amd.tokenizer import pack_sentences

    _, oracle = libs
    p = fixture_dict_parts()
    p["conn_data"] = [0, 100, 200, 100, -30000, 100, 200, 100, -30000]
    p["morphs"] = [[0, 0, 1000], [1, 1, -20000], [2, 2, 1100]]
    d = Dict.from_parts(**p)
    tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
    quirks = ["テ", "テあ", "テ辞書", "テ辞書形態素", "テスト辞書", "ト辞書あ", "辞書テ", "形態素テ形態素", "テテ辞書辞書", ""]
    exp = assert_same(tok, orc, quirks * 60, nthreads=2)  # 600 sentences: beyond the single-launch path
    first_pos = exp.tokens["position"][exp.offsets[:-1][(exp.offsets[1:] - exp.offsets[:-1]) > 0].astype(np.int64)]
    assert (first_pos != 0).any(), "the corpus must contain a chain whose head was dropped"
    # a dictionary word of 5000 characters: its token does not fit kgpu_token8 (chars > 4095)
    p2 = fixture_dict_parts()
    long_word = "あ" * 5000
    kws = sorted(p2["sorted_keywords"] + [long_word], key=lambda s: s.encode())
    morphs = {k: m for k, m in zip(p2["sorted_keywords"], p2["morphs"])}
    morphs[long_word] = [0, 0, -30000]
    p2["sorted_keywords"], p2["morphs"] = kws, [morphs[k] for k in kws]
    d2 = Dict.from_parts(**p2)
    tok2, orc2 = Tokenizer(d2), oracle.OracleTokenizer.from_dict(d2)
    exp2 = assert_same(tok2, orc2, ["テスト"] * 150 + [long_word + "テスト"] + ["辞書"] * 150, nthreads=2)
    assert (exp2.tokens["end"] - exp2.tokens["start"] == 5000).any()
    # the legacy form of a large call gives the same records
    sd, tok3, orc3 = small
    sents = synth.make_corpus(sd, 9000, 61, "cfg2") + [""] + synth.make_corpus(sd, 300, 62, "cfg3")
    a = assert_same(tok3, orc3, sents)
    monkeypatch.setenv("KGPU_HOST_LEGACY", "1")
    b = assert_same(tok3, orc3, sents)
    assert np.array_equal(a.tokens, b.tokens)


def test_large_host_call_chunk_byte_counts(small):
    """Regression: the staging copy of a chunk is split between the calling thread and the workers; a split rounded DOWN lost the
    chunk's last bytes (floor(bytes / pieces) a multiple of 64, bytes not a multiple of pieces).  Chunks whose byte counts sit on
    and around such values, the last sentence of each chunk ending in a distinctive character."""
    sd, tok, orc = small
    for extra in (0, 1, 2, 3, 5, 7):
        body = ["あ" * 85 + "a"] * 2047          # 256 bytes each: chunk 0 = 2048 sentences (the pipeline's chunk for 2049..24576 sentences)
        last = "あ" * 85 + "a" + "z" * extra + "Q"  # 257 + extra bytes: chunk 0 has 2 * 262144 + 1 + extra bytes
        sents = body + [last] + ["い" * 30 + "b"] * 200
        assert sum(len(x.encode()) for x in sents[:2048]) == 2 * 262144 + 1 + extra
        assert_same(tok, orc, sents)


@pytest.mark.parametrize("window_kib,pool", [("12", "40:4:40"), ("16", "40:4:48"), ("16", "0"), ("24", "16:4:8")])
def test_windowed_long_sentence_kernel(libs, window_kib, pool, monkeypatch):
    """The windowed kernel (kgpu_window.hip, KGPU_WINDOW = KiB of LDS per workgroup; in the chain for sentences of 3072 bytes and more, for
    everything when there is no pool kernel): the lattice is built and relaxed window by window, only the carry list / the far FIFO /
    16 bytes per node outlive a window (src/lattice.rs:101-154).  What it cannot hold travels on to the general (HBM-scratch) kernel -- same
    records either way."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    monkeypatch.setenv("KGPU_WINDOW", window_kib)
    monkeypatch.setenv("KGPU_POOL", pool)
    sd = synth.build_dict(20000, seed=11)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    sents = (synth.make_corpus(sd, 1500, 3, "cfg2") + synth.make_corpus(sd, 1200, 4, "cfg3") + synth.make_corpus(sd, 6, 6, "cfg5")
             + ["", "あ", "ア" * 1500, "1" * 1025, "𠮷野家" * 40, "漢" * 300, "x" * 1023 + "あ"])
    for _ in range(2):
        assert_same(tok, orc, sents)
    assert_same(tok, orc, ["", "", "すもも", ""])


@pytest.mark.parametrize("window_kib", ["12", "9", "24"])
def test_windowed_kernel_big_buckets_and_many_prefixes(libs, window_kib, monkeypatch):
    """Two things the windowed kernel used to hand on to the general kernel and now holds itself (round 4):
    * a position whose bucket has more than 64 predecessors -- the end of a long same-category run: every start position of the run contributes its
      unknown words there (src/lattice.rs:66-84) -- gets no pair table; its step loads the connection costs itself (connection.rs:12-14);
    * a start position with more than eight dictionary prefixes parks the ninth and later ones in the window's shared overflow area
      (trie/da.rs:155-182 yields them in ascending length; the node order of src/lattice.rs:105-110 must survive).
    Pool off: every sentence goes through the windowed kernel.  Routing must show that nothing was handed on."""
    from kanpyo_amd import Dict, Tokenizer, synth
    from kanpyo_amd.device import PROFILE_OFF

    _, oracle = libs
    monkeypatch.setenv("KGPU_POOL", "0")
    monkeypatch.setenv("KGPU_WINDOW", window_kib)
    # (1) IPADIC-shaped dictionary: katakana / digit / latin runs of 20..200 characters between ordinary words
    sd = synth.build_dict(20000, seed=11)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    rng = np.random.default_rng(int(window_kib))
    words = synth.make_corpus(sd, 400, 7, "cfg2")
    sents = []
    for k in range(300):
        run = "".join(map(chr, rng.integers(0x30A1, 0x30F7, size=int(rng.integers(20, 200))))) if k % 3 else "".join(map(chr, rng.integers(0x30, 0x3A, size=int(rng.integers(20, 200)))))
        sents.append(words[k][: int(rng.integers(0, 30))] + run + words[k + 1][: int(rng.integers(1, 30))] + (run[:40] if k % 5 == 0 else ""))
    ctx, t, toff, utf8, offs = _device_run(tok, sents, PROFILE_OFF)
    exp = orc.tokenize_batch(utf8, offs, 8)
    assert np.array_equal(toff.astype(np.uint64), exp.offsets) and np.array_equal(t.reshape(-1), exp.tokens.view(np.int32).reshape(-1))
    if window_kib != "9":   # (at 9 KB some windows do not fit even four positions long: those sentences may travel on -- still the same records)
        assert ctx.profile()["deferred"][0] == 0, ctx.profile()
    # (2) nested prefixes: "あ", "ああ", ... up to 30 characters, three records each: up to 30 prefixes per start position
    kws = []
    for n in range(1, 31):
        kws += ["あ" * n] * 3
    kws += ["い", "あい"]
    kws = sorted(kws, key=lambda x: x.encode())
    nr = np.random.default_rng(3)
    morphs = np.stack([nr.integers(0, 5, len(kws)), nr.integers(0, 5, len(kws)), nr.integers(-500, 5000, len(kws))], axis=1)
    p = fixture_dict_parts()
    d = Dict.from_parts(kws, morphs, 5, 5, nr.integers(-900, 900, 25), p["char_class"], p["char_category"], p["invoke_list"], p["group_list"],
                        {0: (1, 1), 1: (1, 2), 2: (2, 1)}, [[0, 0, 4000], [1, 1, 3500]])
    tok2, orc2 = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
    sents2 = ["あ" * n for n in (1, 8, 9, 10, 17, 33, 64, 200)] + ["あ" * 12 + "い" + "あ" * 40, "いあいあ" * 30, "あ" * 5 + "x" + "あ" * 70]
    ctx2, t2, toff2, utf82, offs2 = _device_run(tok2, sents2, PROFILE_OFF)
    exp2 = orc2.tokenize_batch(utf82, offs2, 2)
    assert np.array_equal(toff2.astype(np.uint64), exp2.offsets) and np.array_equal(t2.reshape(-1), exp2.tokens.view(np.int32).reshape(-1))


@pytest.mark.parametrize("pool,window_kib", [("40:4:40", "12"), ("0", "12"), ("0", "0")])
def test_dictionary_keys_of_every_utf8_width(libs, pool, window_kib, monkeypatch):
    """The device walks a character-level copy of the trie (kgpu_chartrie.cpp; reference walk: trie/da.rs:155-182, byte by byte).  Keys of 1-,
    2-, 3- and 4-byte characters, keys that are prefixes of each other across widths, characters beyond the kernels' BMP table (non-BMP, and
    U+FFFF, which shares the table's "not here" value) at the start, in the middle and at the end of keys -- through the pool kernel, the
    windowed kernel and the general kernel (pool off / window off), short and very long sentences, the single-launch small-call path."""
    from kanpyo_amd import Dict, Tokenizer

    _, oracle = libs
    monkeypatch.setenv("KGPU_POOL", pool)
    monkeypatch.setenv("KGPU_WINDOW", window_kib)
    rng = np.random.default_rng(17)
    words = ["a", "ab", "abc", "é", "éa", "aé", "あ", "あい", "あ𠮷", "𠮷", "𠮷野", "𠮷野家", "野家", "😀", "😀😀", "a😀b", "￿", "￿x", "x￿y",
             "い￿", "東京", "東京都", "京都", "都", "xyz𩸽", "𩸽", "λ", "λμ", "μあλ"]
    kws = []
    for w in sorted(set(words), key=lambda s: s.encode()):
        kws += [w] * int(rng.integers(1, 4))
    morphs = np.stack([rng.integers(0, 6, len(kws)), rng.integers(0, 6, len(kws)), rng.integers(-300, 6000, len(kws))], axis=1)
    p = fixture_dict_parts()
    d = Dict.from_parts(kws, morphs, 6, 6, rng.integers(-900, 900, 36), p["char_class"], p["char_category"],
                        p["invoke_list"], p["group_list"], {0: (1, 1), 1: (1, 2), 2: (2, 1)}, [[0, 0, 4000], [1, 1, 3500]])
    tok, orc = Tokenizer(d), oracle.OracleTokenizer.from_dict(d)
    alphabet = list("abcéあい𠮷野家😀￿xy東京都𩸽λμz ")
    sents = ["".join(rng.choice(alphabet, size=int(n))) for n in rng.integers(1, 60, 400)]
    sents += ["".join(rng.choice(alphabet, size=int(n))) for n in (700, 1500, 2500)]  # well beyond the pool kernel's reach: the windowed kernel
    sents += ["", "𠮷", "￿", "😀" * 300, "𠮷野家" * 500, "a￿x￿y" * 200] + words
    assert_same(tok, orc, sents)
    assert_same(tok, orc, sents[:5])  # small call


def test_byte_level_walk_is_kept_and_agrees(small, monkeypatch):
    """KGPU_BYTE_TRIE=1: no character-level copy of the trie, every kernel walks the reference's byte-level double array as in rounds 1-2 (the
    path a dictionary with 65535 or more distinct characters takes, and the one the work counters' byte steps come from)."""
    from kanpyo_amd import Tokenizer, synth

    sd, _, orc = small
    monkeypatch.setenv("KGPU_BYTE_TRIE", "1")
    tok = Tokenizer(sd.dict)
    monkeypatch.delenv("KGPU_BYTE_TRIE")
    assert_same(tok, orc, synth.make_corpus(sd, 3000, 41, "cfg2") + synth.make_corpus(sd, 300, 42, "cfg3") + synth.make_corpus(sd, 3, 43, "cfg5"))
    assert_same(tok, orc, synth.make_corpus(sd, 5, 44, "cfg2"))


def test_chain_tail_is_left_out_and_comes_back(libs):
    """While no recent batch left a sentence for the windowed kernel, the launch chain ends behind the pool kernel (no empty tail launch);
    a batch that does need the tail is found by the last work list's count and run again with it (kgpu_routing.tail_reruns) -- same records
    (src/tokenizer.rs:16: calls are independent, a rerun is invisible)."""
    from kanpyo_amd import Tokenizer, synth
    from kanpyo_amd.device import PROFILE_OFF

    _, oracle = libs
    sd = synth.build_dict(20000, seed=5)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    short = [s[:30] for s in synth.make_corpus(sd, 600, 9, "cfg2")]
    reruns = 0
    for k in range(14):  # the dictionary's contexts start with the tail armed; eight clean batches disarm it
        ctx, t, toff, utf8, offs = _device_run(tok, short, PROFILE_OFF)
        reruns += ctx.profile()["tail_reruns"]
    assert reruns == 0
    mixed = short[:100] + ["ア" * 900, "漢字かな" * 150] + synth.make_corpus(sd, 5, 10, "cfg3") + short[100:200]
    ctx, t, toff, utf8, offs = _device_run(tok, mixed, PROFILE_OFF)
    exp = orc.tokenize_batch(utf8, offs, 8)
    assert np.array_equal(toff.astype(np.uint64), exp.offsets)
    assert np.array_equal(t.reshape(-1), exp.tokens.view(np.int32).reshape(-1))
    assert ctx.profile()["tail_reruns"] == 1
    ctx, t, toff, utf8, offs = _device_run(tok, mixed, PROFILE_OFF)  # armed again: no rerun
    assert np.array_equal(t.reshape(-1), exp.tokens.view(np.int32).reshape(-1)) and ctx.profile()["tail_reruns"] == 0


def test_long_batches_start_with_the_windowed_kernel(libs, monkeypatch):
    """A batch whose sentences average >= KGPU_WINDOW_FIRST bytes (default 1024: the host knows n and the byte count, not the lengths) gets no pool launch
    in front -- the windowed kernel takes the whole batch, short sentences included, on one of the dictionary's long streams -- and the records are the
    same as through the pool-first chain and the oracle's (src/tokenizer.rs:16: a call's result does not depend on how it was scheduled).  Alternating
    long and short batches on ONE context switches its stream back and forth; the host-buffer entry point takes the same decision per chunk."""
    from kanpyo_amd import Tokenizer, synth
    from kanpyo_amd.device import PROFILE_OFF
    from kanpyo_amd.tokenizer import pack_sentences

    _, oracle = libs
    sd = synth.build_dict(20000, seed=5)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    docs = synth.make_corpus(sd, 40, 7, "cfg5") + synth.make_corpus(sd, 30, 8, "cfg2") + ["", "ア" * 1500, "あ"]   # average ~2.7 KB: window first
    short = synth.make_corpus(sd, 700, 9, "cfg2")
    utf8, offs = pack_sentences(docs)
    assert int(offs[-1]) >= 1024 * len(docs)
    exp = orc.tokenize_batch(utf8, offs, 8)
    ctx, t, toff, _, _ = _device_run(tok, docs, PROFILE_OFF)
    plan, prof = ctx.plan(), ctx.profile()
    assert plan["window_first_bytes"] == 1024
    assert np.array_equal(toff.astype(np.uint64), exp.offsets) and np.array_equal(t.reshape(-1), exp.tokens.view(np.int32).reshape(-1))
    assert prof["long_launches"] == 1 and prof["deferred"][0] == 0 and prof["redone"][0] == 0   # list 0 is the windowed kernel's own output: nothing left, nothing routed
    # the same context, alternating: short batch (pool first, shared stream), long, short
    import torch

    from kanpyo_amd.device import DeviceContext

    dev = torch.device("cuda", 0)
    c = DeviceContext(tok)
    for sents in (short, docs, short, docs):
        u, o = pack_sentences(sents)
        e = orc.tokenize_batch(u, o, 8)
        n, cap = len(sents), int(o[-1]) + len(sents)
        bufs = (torch.from_numpy(u.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), torch.empty((cap, 6), dtype=torch.int32, device=dev),
                torch.empty(n + 1, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
        c.tokenize(bufs[0].data_ptr(), bufs[1].data_ptr(), n, int(o[-1]), bufs[2].data_ptr(), cap, bufs[3].data_ptr(), bufs[4].data_ptr())
        nt = c.sync()
        assert np.array_equal(bufs[3].cpu().numpy().astype(np.uint64), e.offsets)
        assert np.array_equal(bufs[2][:nt].cpu().numpy().reshape(-1), e.tokens.view(np.int32).reshape(-1))
    # the host-buffer entry point (chunks of its pipeline take the decision one by one)
    assert_same(tok, orc, docs * 3 + short + docs)
    # with the rule switched off the pool kernel routes as before, and with the two-wavefronts-per-sentence form forced on (or off) for the window-first
    # chain the records are the same again (the knobs are read with the launch plan, when a context is created)
    for env, need_deferred in (({"KGPU_WINDOW_FIRST": "0"}, True), ({"KGPU_WINDOW_TEAM": "2"}, False), ({"KGPU_WINDOW_TEAM": "0"}, False)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tok2 = Tokenizer(sd.dict)
        t2, toff2, st2 = tok2.tokenize_packed(utf8, offs)
        assert np.array_equal(toff2, exp.offsets) and np.array_equal(t2, exp.tokens) and not st2.any(), env
        r = tok2.routing()
        assert (r["deferred"][0] >= 40) == need_deferred, (env, r)
        tok2.close()
        for k in env:
            monkeypatch.delenv(k)


def test_callers_own_stream_is_never_left(libs):
    """A context created on the caller's stream keeps every launch on it -- window-first chains, the team form, chains with a large windowed share included (the
    long streams are for contexts on the library's own streams): work the caller queues on that stream before and after a batch stays ordered with it."""
    import torch

    from kanpyo_amd import Tokenizer, synth
    from kanpyo_amd.device import DeviceContext
    from kanpyo_amd.tokenizer import pack_sentences

    _, oracle = libs
    sd = synth.build_dict(20000, seed=5)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    c = DeviceContext(tok, st.cuda_stream)
    assert c.plan()["long_streams"] == 0
    docs = synth.make_corpus(sd, 30, 7, "cfg5")
    mix = synth.make_corpus(sd, 900, 8, "cfg3")
    short = synth.make_corpus(sd, 500, 9, "cfg2")
    for sents in (docs, mix, mix, short, docs):
        u, o = pack_sentences(sents)
        e = orc.tokenize_batch(u, o, 8)
        n, cap = len(sents), int(o[-1]) + len(sents)
        with torch.cuda.stream(st):   # the input is produced on the caller's stream, right in front of the batch: no synchronisation in between
            d_u = torch.from_numpy(u.copy()).to(dev, non_blocking=False)
            d_o = torch.from_numpy(o.astype(np.int64)).to(dev, non_blocking=False)
            d_t = torch.empty((cap, 6), dtype=torch.int32, device=dev)
            d_toff = torch.empty(n + 1, dtype=torch.int64, device=dev)
            d_st = torch.empty(n, dtype=torch.uint8, device=dev)
            c.tokenize(d_u.data_ptr(), d_o.data_ptr(), n, int(o[-1]), d_t.data_ptr(), cap, d_toff.data_ptr(), d_st.data_ptr())
            total = d_toff[-1:].clone()   # queued on the same stream BEHIND the batch: must see its result without any host synchronisation
        nt = c.sync()
        assert nt == len(e.tokens)
        p = c.profile()
        if p["tail_reruns"] == 0 and p["arena_regrows"] == 0 and p["window_reruns"] == 0:   # (a batch the host had to complete at sync time is final only then)
            assert int(total.cpu()[0]) == nt
        assert np.array_equal(d_toff.cpu().numpy().astype(np.uint64), e.offsets)
        assert np.array_equal(d_t[:nt].cpu().numpy().reshape(-1), e.tokens.view(np.int32).reshape(-1))
    c.close()
    tok.close()

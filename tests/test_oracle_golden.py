"""Pin the CPU oracle (C restatement + pure-Python restatement) against every
known-answer vector the reference's own tests hold for the hot path
(SURVEY.md 8c), and against the hand-derived App. C token vectors.  CPU only."""
import numpy as np
import pytest

from conftest import fixture_dict_parts, load_golden
from oracle import oracle, pyref


def _blobs_for_keywords(keywords, n_morph_pad=None):
    """A minimal dict around an index built by the ORACLE's own builder."""
    n = n_morph_pad or max(len(keywords), 1)
    import struct

    index = oracle.index_build(keywords)
    morph = struct.pack("<q", n) + np.zeros((n, 3), dtype="<i2").tobytes()
    conn = struct.pack("<QQ", 1, 1) + np.zeros(1, dtype="<i2").tobytes()
    unk = struct.pack("<Q", 0) + struct.pack("<q", 0)
    cat = np.zeros(65536, dtype=np.uint8)
    return index, conn, morph, unk, cat, np.zeros(1, np.uint8), np.zeros(1, np.uint8)


def test_da_search_ascii():
    g = load_golden("trie_kat.json")["search_ascii"]
    idx = oracle.index_build(g["keywords"])
    for i, k in enumerate(g["keywords"]):
        assert oracle.da_search(idx, k) == i + 1
    for k in g["not_found"]:
        assert oracle.da_search(idx, k) is None


def test_da_search_multibyte():
    g = load_golden("trie_kat.json")["search_multibyte"]
    kws = sorted(g["keywords_unsorted"], key=lambda s: s.encode("utf-8"))
    idx = oracle.index_build(kws)
    for i, k in enumerate(kws):
        assert oracle.da_search(idx, k) == i + 1
    for k in g["not_found"]:
        assert oracle.da_search(idx, k) is None


def test_common_prefix_exact_lists():
    g = load_golden("trie_kat.json")["common_prefix"]
    parts = _blobs_for_keywords(g["keywords"])
    o = oracle.OracleTokenizer(*parts)
    p = pyref.PyDict(*parts)
    for q in g["queries"]:
        exp = None if q["expect"] is None else [tuple(x) for x in q["expect"]]
        assert o.common_prefix(q["text"]) == exp
        assert p.search_common_prefix_of(q["text"].encode()) == exp


def test_index_duplicates_and_prefix_counts():
    g = load_golden("trie_kat.json")
    d = g["index_dups"]
    o = oracle.OracleTokenizer(*_blobs_for_keywords(d["keywords"]))
    for k, c in d["counts"].items():
        r = o.common_prefix(k)
        assert r is not None and len(r) == c
        assert [x[0] for x in r] == list(range(r[0][0], r[0][0] + c))  # id..=id+dup (index.rs:48-50)
    assert o.common_prefix(d["not_found"]) is None
    nf = g["index_not_found"]
    assert oracle.OracleTokenizer(*_blobs_for_keywords(nf["keywords"])).common_prefix(nf["query"]) is None
    pf = g["index_prefixes"]
    r = oracle.OracleTokenizer(*_blobs_for_keywords(pf["keywords"])).common_prefix(pf["query"])
    assert len(r) == pf["count"]


def test_index_build_empty():
    # index.rs:92-95: building from an empty keyword list succeeds
    blob = oracle.index_build([])
    assert len(blob) == 8 + 8 + 8  # 1-node array (truncate keeps slot 0) + empty dup map


def test_connection_layout():
    g = load_golden("matrix_kat.json")["get"]
    import struct

    # get(i, j) == data[j*row + i]: a 2-morph dict whose costs expose the element picked
    conn = struct.pack("<QQ", g["row"], g["col"]) + np.array(g["data"], dtype="<i2").tobytes()
    index = oracle.index_build(["a", "b"])
    for li in range(2):
        for ri in range(2):
            # morph 'a' has right_id = ri, morph 'b' has left_id = li; cost of path a,b includes M[ri][li]
            morph = struct.pack("<q", 2) + np.array([[0, ri, 0], [li, 0, 0]], dtype="<i2").tobytes()
            unk = struct.pack("<Q", 0) + struct.pack("<q", 0)
            cat = np.zeros(128, dtype=np.uint8)
            parts = (index, conn, morph, unk, cat, np.zeros(1, np.uint8), np.zeros(1, np.uint8))
            toks, ctr = oracle.OracleTokenizer(*parts).tokenize("ab")
            assert [int(t["id"]) for t in toks] == [1, 2, 0]
            assert pyref.tokenize(pyref.PyDict(*parts), "ab")[0][0] == 1
    # direct statement of connection.rs:58-71
    row = g["row"]
    for i in range(g["row"]):
        for j in range(g["col"]):
            assert g["data"][j * row + i] == j * row + i


def _fixture_oracles():
    from kanpyo_amd.dict import connection_blob, morphs_blob, unk_blob

    p = fixture_dict_parts()
    parts = (
        oracle.index_build(p["sorted_keywords"]), connection_blob(p["conn_rows"], p["conn_cols"], p["conn_data"]),
        morphs_blob(p["morphs"]), unk_blob(p["unk_map"], p["unk_morphs"]), p["char_category"], p["invoke_list"],
        p["group_list"],
    )
    return oracle.OracleTokenizer(*parts), pyref.PyDict(*parts)


def _as_tuples(text, toks):
    raw = text.encode("utf-8")
    out = []
    for t in toks:
        pos, bl, cls = int(t["position"]), int(t["byte_len"]), int(t["cls"])
        surface = "EOS" if cls == 0 else raw[pos : pos + bl].decode("utf-8")
        out.append([int(t["id"]), cls, pos, int(t["start"]), int(t["end"]), surface])
    return out


def test_fixture_tokens_app_c():
    """App. C vectors: C oracle == pure-Python restatement == hand derivation."""
    o, p = _fixture_oracles()
    for case in load_golden("fixture_tokens.json")["cases"]:
        toks, ctr = o.tokenize(case["input"])
        assert _as_tuples(case["input"], toks) == case["tokens"], case["input"]
        py = pyref.tokenize(p, case["input"])
        raw = case["input"].encode()
        assert [[i, c, pos, s, e, "EOS" if c == 0 else raw[pos : pos + bl].decode()] for i, c, pos, s, e, bl in py] == case["tokens"]


def test_fixture_reference_assertions():
    """The properties the reference itself asserts (src/tests.rs:110-171)."""
    o, _ = _fixture_oracles()
    toks, _ = o.tokenize("テスト")
    assert len(toks) > 0 and any(int(t["cls"]) != 0 for t in toks)
    for t in toks:
        if int(t["cls"]) != 0:
            assert t["start"] <= t["end"] and t["end"] <= len("テスト")
    assert len(o.tokenize("")[0]) > 0
    assert len(o.tokenize("あいうえお")[0]) > 0


def test_counters_on_fixture():
    o, _ = _fixture_oracles()
    _, c = o.tokenize("辞書形態素")
    # 9 nodes incl. BOS -> N excludes BOS
    assert c["N"] == 8 and c["C"] == 5 and c["B"] == 15 and c["K"] == 3


def test_invalid_utf8_is_rejected():
    o, _ = _fixture_oracles()
    for bad in (b"\xff", b"\xe3\x81", b"\x80", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"a\xe3\x81\x82\xe3"):
        with pytest.raises(UnicodeDecodeError):
            o.tokenize(bad)


def test_slots_form_equals_dense_form(oracle_mod):
    """korc_tokenize_slots (the all-core timing form of bench.py's cpu_baseline: per-sentence slots, claimed runs,
    several passes) yields the same records as the dense single-thread batch."""
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd = synth.build_dict(6000, seed=3)
    orc = oracle_mod.OracleTokenizer.from_dict(sd.dict)
    sents = synth.make_corpus(sd, 700, 5, "cfg2") + ["", "あ"] + synth.make_corpus(sd, 50, 6, "cfg3")
    utf8, offs = pack_sentences(sents)
    exp = orc.tokenize_batch(utf8, offs, 1)
    for nthreads, reps in ((1, 1), (4, 1), (7, 3)):
        slots, cnt, ctr = orc.tokenize_slots(utf8, offs, nthreads, reps)
        assert np.array_equal(cnt.astype(np.uint64), exp.offsets[1:] - exp.offsets[:-1])
        for i in (0, 1, 350, 699, 700, 701, len(sents) - 1):
            lo = int(offs[i]) + i
            assert np.array_equal(slots[lo : lo + int(cnt[i])], exp.tokens[int(exp.offsets[i]) : int(exp.offsets[i + 1])])
        dense = np.concatenate([slots[int(offs[i]) + i : int(offs[i]) + i + int(cnt[i])] for i in range(len(sents))])
        assert np.array_equal(dense, exp.tokens)
        assert ctr["sentences"] == reps * len(sents) and ctr["K"] == reps * exp.counters["K"]

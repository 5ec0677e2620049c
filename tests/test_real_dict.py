"""A real Kanpyo dictionary, when one is there (reference README.md:74-82, build.rs:4-5,27-46: ipa.dict is fetched from GitHub Releases -- it cannot be
obtained in this environment, so these tests are skipped unless the variables are set):

    KANPYO_DICT=/path/ipa.dict               a Kanpyo .dict (zip of the six blobs, kanpyo-dict/src/dict.rs:51-116)
    KANPYO_SENTENCES=/path/sentences.txt     optional: UTF-8 text, one sentence per line (default: a built-in handful + synthetic noise)
    KANPYO_EXPECTED=/path/expected.txt       optional: `kanpyo tokenize < sentences.txt` of the REFERENCE binary -- the one artefact that would pin
                                             token-level parity against the reference itself

With KANPYO_DICT alone: GPU vs oracle on that dictionary, every launch chain.  With KANPYO_EXPECTED too: the GPU's CLI lines equal the reference's.
The `-m "not gpu"` half runs the same loader + oracle path on a .dict this repository writes itself, so the plumbing is exercised without the file."""
import os

import numpy as np
import pytest

DICT = os.environ.get("KANPYO_DICT")
BUILTIN = ["すもももももももものうち", "東京都に住んでいます。", "吾輩は猫である。名前はまだ無い。", "Ｇｏｏｇｌｅで検索する", "１２３４５円のｉＰｈｏｎｅ", "", "ｶﾀｶﾅとひらがなと漢字",
           "メロスは激怒した。必ず、かの邪智暴虐の王を除かなければならぬと決意した。"]


def _sentences():
    p = os.environ.get("KANPYO_SENTENCES")
    if p:
        with open(p, encoding="utf-8") as f:
            lines = [ln.rstrip() for ln in f.read().split("\n")]
        if lines and lines[-1] == "":
            lines.pop()
        return lines
    rng = np.random.default_rng(7)
    pool = "".join(BUILTIN)
    noise = ["".join(rng.choice(list(pool), size=int(rng.integers(1, 400)))) for _ in range(300)]
    return BUILTIN + noise


def _check_against_oracle(df, sents, tokenizer_cls, oracle):
    from kanpyo_amd.tokenizer import pack_sentences

    tok = tokenizer_cls(df.dict)
    orc = oracle.OracleTokenizer.from_dict(df.dict)
    utf8, offs = pack_sentences(sents)
    t, toff, st = tok.tokenize_packed(utf8, offs)
    exp = orc.tokenize_batch(utf8, offs, 8)
    assert not st.any()
    assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens)
    return tok


@pytest.mark.gpu
@pytest.mark.skipif(not DICT, reason="KANPYO_DICT not set: no real dictionary in this environment")
def test_real_dictionary_gpu_vs_oracle_and_reference_output(monkeypatch):
    from kanpyo_amd import Tokenizer
    from kanpyo_amd.dictfile import format_tokens, load_dict
    from oracle import oracle

    oracle.build()
    df = load_dict(DICT)
    sents = _sentences()
    for env in ({}, {"KGPU_POOL": "0"}, {"KGPU_POOL": "0", "KGPU_WINDOW": "0"}, {"KGPU_NO_SMALL_CALLS": "1"}):  # shipped plan; windowed kernel alone; general kernel alone; no single-launch path
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tok = _check_against_oracle(df, sents, Tokenizer, oracle)
        for k in env:
            monkeypatch.delenv(k)
    exp_path = os.environ.get("KANPYO_EXPECTED")
    if exp_path:
        got = []
        for toks in tok.tokenize_batch(sents):
            got += format_tokens(toks, df).split("\n") if toks else []
        with open(exp_path, encoding="utf-8") as f:
            exp = f.read().split("\n")
        if exp and exp[-1] == "":
            exp.pop()
        assert got == exp, "the GPU's `surface\\tfeatures` lines differ from the reference binary's"


def test_the_same_plumbing_on_a_dict_file_written_here(tmp_path, oracle_mod):
    """load_dict -> oracle on a .dict this repository writes (no GPU): what KANPYO_DICT would go through, minus the device."""
    from kanpyo_amd import synth
    from kanpyo_amd.dictfile import DictFile, MorphFeatureTable, load_dict, save_dict
    from kanpyo_amd.tokenizer import pack_sentences

    sd = synth.build_dict(6000, seed=3)
    n = sd.dict.n_morphs
    feats = MorphFeatureTable.from_features([["名詞", "一般", str(i % 97)] for i in range(n)])
    p = tmp_path / "t.dict"
    save_dict(DictFile(sd.dict, feats, MorphFeatureTable.from_features([["未知語"]] * 40)), p)
    df = load_dict(str(p))
    sents = synth.make_corpus(sd, 50, 1, "cfg2")
    utf8, offs = pack_sentences(sents)
    a = oracle_mod.OracleTokenizer.from_dict(df.dict).tokenize_batch(utf8, offs, 2)
    b = oracle_mod.OracleTokenizer.from_dict(sd.dict).tokenize_batch(utf8, offs, 2)
    assert np.array_equal(a.tokens, b.tokens) and np.array_equal(a.offsets, b.offsets)

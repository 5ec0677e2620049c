"""MeCab-source builder: parsers against the reference's known
answers, record sort order => token ids, and an end-to-end build of a tiny
MeCab-format directory checked through the oracle.  CPU only."""
import os

import numpy as np
import pytest

from conftest import load_golden
from kanpyo_amd import builder
from kanpyo_amd.dictfile import format_tokens, load_dict, save_dict
from kanpyo_amd.token import Token, TokenClass

CHAR_DEF = """# test char.def in MeCab format
DEFAULT         0 1 0  # mandatory
SPACE           0 1 0
KANJI           0 0 2
HIRAGANA        0 1 2
KATAKANA        1 1 2
NUMERIC         1 1 0

0x0020 SPACE
0x0030..0x0039 NUMERIC
0x3041..0x309F  HIRAGANA
0x30A1..0x30FF  KATAKANA
0x4E00..0x9FA5  KANJI
0x4E00 NUMERIC KANJI  # first category wins (char_def.rs:65-71)
"""
UNK_DEF = "DEFAULT,5,5,4769,記号,一般,*,*,*,*,*\nKATAKANA,1,1,3000,名詞,一般,*,*,*,*,*\nKATAKANA,1,1,2500,名詞,固有,*,*,*,*,*\nHIRAGANA,2,2,9000,助詞,*,*,*,*,*,*\nKANJI,1,1,8000,名詞,一般,*,*,*,*,*\nNUMERIC,3,3,1000,名詞,数,*,*,*,*,*\nSPACE,4,4,100,記号,空白,*,*,*,*,*\n"
LEX_A = "東京,1,1,3000,名詞,固有名詞,地域,一般,*,*,東京,トウキョウ,トーキョー\n東京都,1,1,3500,名詞,固有名詞,地域,一般,*,*,東京都,トウキョウト,トーキョート\nに,2,2,500,助詞,格助詞,一般,*,*,*,に,ニ,ニ\n"
LEX_B = "住む,6,6,4000,動詞,自立,*,*,五段・マ行,基本形,住む,スム,スム\nに,2,2,700,助詞,副助詞,*,*,*,*,に,ニ,ニ\n都,1,1,6000,名詞,接尾,地域,*,*,*,都,ト,ト\n\"a,b\",1,1,100,名詞,一般,*,*,*,*,\"a,b\",*,*\n"


def _matrix(n=7):
    lines = [f"{n} {n}"]
    for r in range(n):
        for c in range(n):
            lines.append(f"{r} {c} {(r * 31 + c * 17) % 200 - 100}")
    return "\n".join(lines) + "\n"


def test_matrix_def_known_answer():
    g = load_golden("matrix_kat.json")["matrix_def"]
    assert builder.parse_matrix_def(g["text"]) == (2, 2, g["data"])
    with pytest.raises(builder.BuilderError):
        builder.parse_matrix_def("2 2\n0 5 1\n")
    with pytest.raises(builder.BuilderError):
        builder.parse_matrix_def("2 2\n0 0\n")


def test_char_def_parser():
    cc, cat, inv, grp = builder.parse_char_def(CHAR_DEF)
    assert cc == ["DEFAULT", "SPACE", "KANJI", "HIRAGANA", "KATAKANA", "NUMERIC"]
    assert inv.tolist() == [0, 0, 0, 0, 1, 1] and grp.tolist() == [1, 1, 0, 1, 1, 1]
    assert cat.size == 65536 and cat[0x20] == 1 and cat[ord("5")] == 5 and cat[ord("あ")] == 3 and cat[ord("ア")] == 4
    assert cat[ord("東")] == 2 and cat[0x4E00] == 5  # "0x4E00 NUMERIC KANJI": only the first category is used
    assert cat[ord("A")] == 0  # default class 0
    with pytest.raises(builder.BuilderError):
        builder.parse_char_def("DEFAULT 0 1 0\n0x00e9 DEFAULT\n")  # lower-case hex is not matched by the reference regex
    with pytest.raises(builder.BuilderError):
        builder.parse_char_def("DEFAULT 0 1 0\n0x0041 ALPHA\n")    # unknown class: cc2id[..] panics


def test_end_to_end_build_and_ids(tmp_path):
    for name, text in (("a.csv", LEX_A), ("b.csv", LEX_B), ("matrix.def", _matrix()), ("char.def", CHAR_DEF), ("unk.def", UNK_DEF)):
        (tmp_path / name).write_bytes(text.encode("utf-8"))
    df = builder.build_from_dir(str(tmp_path), encoding="utf-8")
    # ids follow the derived Ord of Record: surface bytes, left, right, cost, features (record.rs:5-19)
    recs = sorted(builder.parse_csv(LEX_A) + builder.parse_csv(LEX_B), key=builder._record_key)
    assert [r[0] for r in recs] == ["a,b", "に", "に", "住む", "東京", "東京都", "都"]
    assert [r[3] for r in recs if r[0] == "に"] == [500, 700]
    assert df.dict.n_morphs == 7
    # unknown rows sorted by category name then fields; ids from 1; (first id, count) per category byte
    import struct
    k = struct.unpack_from("<Q", df.dict.unk_dict, 0)[0]
    ents = {struct.unpack_from("<BqQ", df.dict.unk_dict, 8 + 17 * i)[0]: struct.unpack_from("<BqQ", df.dict.unk_dict, 8 + 17 * i)[1:] for i in range(k)}
    assert ents == {0: (1, 1), 3: (2, 1), 2: (3, 1), 4: (4, 2), 5: (6, 1), 1: (7, 1)}  # DEFAULT,HIRAGANA,KANJI,KATAKANA x2,NUMERIC,SPACE
    # tokenise through the oracle (no GPU here) and print like the CLI
    from oracle import oracle, pyref

    text = "東京都に住む"
    toks, _ = oracle.OracleTokenizer.from_dict(df.dict).tokenize(text)
    assert [tuple(x) for x in toks.tolist()] == pyref.tokenize(pyref.PyDict(*[getattr(df.dict, f) for f in (
        "index_dict", "connection_dict", "morph_dict", "unk_dict", "char_category", "invoke_list", "group_list")]), text)
    raw = text.encode()
    tt = [Token(int(t["id"]), TokenClass(int(t["cls"])), int(t["position"]), int(t["start"]), int(t["end"]),
                "EOS" if t["cls"] == 0 else raw[int(t["position"]):int(t["position"]) + int(t["byte_len"])].decode()) for t in toks]
    out = format_tokens(tt, df)
    assert out.endswith("EOS\t") and "".join(t.surface for t in tt[:-1]) == text
    assert out.splitlines()[0].split("\t")[1].startswith("名詞,固有名詞")
    # and the .dict container round trip keeps everything
    p = tmp_path / "t.dict"
    save_dict(df, str(p))
    back = load_dict(str(p))
    assert back.dict.index_dict == df.dict.index_dict and back.morph_feature_table == df.morph_feature_table


def test_cost_overflow_is_an_error():
    with pytest.raises(builder.BuilderError):
        builder.build([("あ", 1, 1, 40000, ["x"])], _matrix(), CHAR_DEF, [])


def test_euc_jp_decoder_is_the_whatwg_one():
    """encoding_rs::EUC_JP (builder/record.rs:21-26) is the WHATWG decoder: JIS X 0208 through the Windows-31J table.
    Python's own 'euc_jp' codec differs in exactly these cells -- and surfaces decide the record order, hence every id."""
    cells = {b"\xa1\xc1": "～", b"\xa1\xc2": "∥", b"\xa1\xdd": "－", b"\xa1\xf1": "￠", b"\xa1\xf2": "￡",
             b"\xa2\xcc": "￢", b"\xa1\xc0": "＼", b"\xad\xa1": "①", b"\x8e\xb1": "ｱ"}
    for raw, ch in cells.items():
        assert builder.decode_euc_jp(raw) == ch
    assert b"\xa1\xc1".decode("euc_jp") == "〜"  # the codec this module must NOT use
    text = "東京都に住む、カタカナ123abc"
    assert builder.decode_euc_jp(text.encode("euc_jp")) == text
    for bad in (b"\xa1", b"\xff\xa1", b"\x8e\x41", b"\xa9\xa1"):  # truncated, invalid lead, bad kana trail, unassigned cell
        with pytest.raises(builder.BuilderError):
            builder.decode_euc_jp(bad)


def test_build_from_euc_jp_directory(tmp_path):
    """`ipa_dict_builder` reads EUC-JP sources (bin/ipa_dict_builder.rs:38-59): the same directory in EUC-JP and in UTF-8
    gives the same dictionary, with a WAVE DASH cell (0xA1C1) landing on U+FF5E as in the reference."""
    lex = LEX_A + "～,1,1,900,記号,一般,*,*,*,*,～,～,～\n"
    euc, utf = tmp_path / "euc", tmp_path / "utf"
    euc.mkdir(); utf.mkdir()
    enc = lambda t: b"".join(b"\xa1\xc1" if ch == "～" else ch.encode("euc_jp") for ch in t)  # noqa: E731
    for name, text in (("a.csv", lex), ("b.csv", LEX_B), ("char.def", CHAR_DEF), ("unk.def", UNK_DEF)):
        (euc / name).write_bytes(enc(text))
        (utf / name).write_bytes(text.encode("utf-8"))
    for d in (euc, utf):
        (d / "matrix.def").write_bytes(_matrix().encode())
    a, b = builder.build_from_dir(str(euc)), builder.build_from_dir(str(utf), encoding="utf-8")
    assert a.dict.index_dict == b.dict.index_dict and a.dict.morph_dict == b.dict.morph_dict and a.dict.unk_dict == b.dict.unk_dict
    assert a.morph_feature_table == b.morph_feature_table
    from oracle import oracle
    toks, _ = oracle.OracleTokenizer.from_dict(a.dict).tokenize("東京～")
    assert [int(t["cls"]) for t in toks] == [1, 1, 0]  # both known: the ～ record is found under U+FF5E


def test_ragged_csv_is_rejected():
    with pytest.raises(builder.BuilderError):  # csv::Reader with flexible = false (record.rs:27-31)
        builder.parse_csv("あ,1,1,10,x,y\nい,1,1,10,x\n")

"""`.dict` container: bincode-2 varint codec, MorphFeatureTable
builder against the reference's known answers, zip round trip of the reference's
fixture dictionary (src/tests.rs:173-202 does the same round trip), CLI-style lines."""
import io

import numpy as np
import pytest

from conftest import fixture_dict_parts, load_golden
from kanpyo_amd.dict import Dict
from kanpyo_amd.dictfile import (DictFile, MorphFeatureTable, _Reader, decode_chardef, enc_varint, encode_chardef,
                                 format_tokens, load_dict, save_dict)
from kanpyo_amd.token import Token, TokenClass


def test_bincode_varint_known_bytes():
    # bincode 2 spec, "varint" integer encoding of config::standard()
    assert enc_varint(0) == b"\x00" and enc_varint(250) == b"\xfa"
    assert enc_varint(251) == b"\xfb\xfb\x00" and enc_varint(65535) == b"\xfb\xff\xff"
    assert enc_varint(65536) == b"\xfc\x00\x00\x01\x00"
    assert enc_varint(1 << 32) == b"\xfd" + (1 << 32).to_bytes(8, "little")
    for v in (0, 1, 250, 251, 300, 65535, 65536, 70000, 2**32 - 1, 2**32, 2**63):
        r = _Reader(enc_varint(v))
        assert r.varint() == v and r.at == len(enc_varint(v))
    with pytest.raises(ValueError):
        _Reader(b"\xff").varint()


def test_feature_table_builder_known_answers():
    g = load_golden("morph_feature_kat.json")
    t = MorphFeatureTable.from_features([r["input"] for r in g["push"]["rows"]])
    assert t.morph_features == [r["ids"] for r in g["push"]["rows"]]
    assert t.name_list[0] == ""
    rows = g["list"]["rows"]
    t2 = MorphFeatureTable.from_features(rows)
    for i, want in enumerate(rows):
        assert [t2.name_list[j] for j in t2.morph_features[i]] == want
        assert t2.features(i + 1) == want
    back, at = MorphFeatureTable.decode(t2.encode())
    assert back == t2 and at == len(t2.encode())
    # layout: Vec<Vec<u32>> then Vec<String>, lengths and u32s as varints
    assert MorphFeatureTable([[1, 300]], ["", "a"]).encode() == b"\x01\x02\x01\xfb\x2c\x01\x02\x00\x01a"


def test_chardef_roundtrip():
    # the reference's own round-trip fixture (char_category_def.rs:64-81)
    blob = encode_chardef(["class1", "class2", "class3"], np.frombuffer(b"abc", dtype=np.uint8), [True, False, True], [False, True, False])
    assert blob == b"\x03\x06class1\x06class2\x06class3\x03abc\x03\x01\x00\x01\x03\x00\x01\x00"
    cc, cat, inv, grp = decode_chardef(blob)
    assert cc == ["class1", "class2", "class3"] and cat.tobytes() == b"abc" and inv.tolist() == [1, 0, 1] and grp.tolist() == [0, 1, 0]


def _fixture_file() -> DictFile:
    d = Dict.from_parts(**fixture_dict_parts())
    feats = [["名詞", "一般", "*", "*", "*", "*", s, r, r] for s, r in (("テスト", "テスト"), ("辞書", "ジショ"), ("形態素", "ケイタイソ"))]
    unk = [["未知語", "*", "*", "*", "*", "*", "*", "*", "*"]] * 2
    return DictFile(d, MorphFeatureTable.from_features(feats), MorphFeatureTable.from_features(unk))


def test_dict_zip_roundtrip_like_the_reference_test():
    df = _fixture_file()
    buf = io.BytesIO()
    save_dict(df, buf)
    import zipfile

    assert zipfile.ZipFile(io.BytesIO(buf.getvalue())).namelist() == [
        "morph.dict", "morph_feature.dict", "connection.dict", "index.dict", "chardef.dict", "unk.dict"]
    back = load_dict(buf.getvalue())
    a, b = df.dict, back.dict
    assert (a.index_dict, a.connection_dict, a.morph_dict, a.unk_dict) == (b.index_dict, b.connection_dict, b.morph_dict, b.unk_dict)
    assert np.array_equal(a.char_category, b.char_category) and np.array_equal(a.invoke_list, b.invoke_list)
    assert np.array_equal(a.group_list, b.group_list) and list(b.char_class) == ["DEFAULT", "KANJI", "HIRAGANA"]
    assert back.morph_feature_table == df.morph_feature_table and back.unk_feature_table == df.unk_feature_table
    # src/tests.rs:173-202: both dictionaries tokenize "テスト" identically (here: via the oracle, no GPU)
    from oracle import oracle

    t1, _ = oracle.OracleTokenizer.from_dict(a).tokenize("テスト")
    t2, _ = oracle.OracleTokenizer.from_dict(b).tokenize("テスト")
    assert np.array_equal(t1, t2) and len(t1) == 2


def test_cli_style_lines():
    df = _fixture_file()
    toks = [Token(2, TokenClass.Known, 0, 0, 2, "辞書"), Token(1, TokenClass.Unknown, 6, 2, 3, "漢"), Token(0, TokenClass.Dummy, 9, 3, 6, "EOS")]
    assert format_tokens(toks, df) == "辞書\t名詞,一般,*,*,*,*,辞書,ジショ,ジショ\n漢\t未知語,*,*,*,*,*,*,*,*\nEOS\t"

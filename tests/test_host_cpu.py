"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/kanpyo_gpu.h declares, the host-side IndexTable builder is
byte-identical to the oracle's restatement of the reference builder, and
dictionary validation turns reference panics into KGPU_ERR_BAD_DICT.  No compute
calls: there is no GPU here."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT, fixture_dict_parts, load_golden
from kanpyo_amd import _lib
from kanpyo_amd.dict import Dict, connection_blob, index_table_build, morphs_blob, unk_blob
from oracle import oracle


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "kanpyo_gpu.h"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(kgpu_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_token_record_is_24_bytes():
    from kanpyo_amd.tokenizer import TOKEN_DTYPE

    assert TOKEN_DTYPE.itemsize == 24 and oracle.TOKEN_DTYPE == TOKEN_DTYPE


@pytest.mark.parametrize("name", ["search_ascii", "common_prefix", "index_dups", "index_prefixes"])
def test_index_build_byte_identical_to_oracle_builder(name):
    kws = load_golden("trie_kat.json")[name]["keywords"]
    assert index_table_build(kws) == oracle.index_build(kws)


def test_index_build_byte_identical_on_synthetic_lexicon():
    from kanpyo_amd import synth

    sd = synth.build_dict(20000, seed=11)
    # the product built sd.dict.index_dict; rebuild the same keyword list with the oracle's builder
    rng = np.random.default_rng(5)
    kws = sorted({"".join(map(chr, rng.integers(0x3041, 0x3060, size=rng.integers(1, 6)))).encode() for _ in range(5000)})
    kws = kws + kws[:50]
    kws.sort()
    assert index_table_build(kws) == oracle.index_build(kws)
    assert sd.dict.da_len > 1000


def test_index_build_empty_and_unsorted():
    assert index_table_build([]) == oracle.index_build([])
    # the reference's own fixture (src/tests.rs:9-13) is NOT byte-sorted; unsorted
    # but prefix-grouped input must build, identically to the reference's recursion
    kws = load_golden("fixture_dict.json")["sorted_keywords"]
    assert [k.encode() for k in kws] != sorted(k.encode() for k in kws)
    assert index_table_build(kws) == oracle.index_build(kws)
    assert index_table_build(["b", "a"]) == oracle.index_build(["b", "a"])
    # same byte in two separate runs: the reference asserts (da.rs:106-111)
    with pytest.raises(_lib.KgpuError) as e:
        index_table_build(["ab", "b", "ac"])
    assert e.value.code == _lib.KGPU_ERR_INVALID_ARG
    with pytest.raises(RuntimeError):
        oracle.index_build(["ab", "b", "ac"])


def test_da_search_known_answers_via_product_builder():
    g = load_golden("trie_kat.json")["search_ascii"]
    idx = index_table_build(g["keywords"])
    for i, k in enumerate(g["keywords"]):
        assert oracle.da_search(idx, k) == i + 1
    for k in g["not_found"]:
        assert oracle.da_search(idx, k) is None


def test_blob_layouts():
    assert morphs_blob([[1, 2, 3]]) == struct.pack("<qhhh", 1, 1, 2, 3)
    assert connection_blob(2, 2, [0, 1, 2, 3]) == struct.pack("<QQhhhh", 2, 2, 0, 1, 2, 3)
    assert unk_blob({2: (2, 1), 1: (1, 1)}, [[0, 0, 5], [1, 1, 6]]) == (
        struct.pack("<Q", 2) + struct.pack("<BqQ", 1, 1, 1) + struct.pack("<BqQ", 2, 2, 1) + morphs_blob([[0, 0, 5], [1, 1, 6]])
    )


def _create(d: Dict):
    b = _lib.DictBlobs()
    keep = []
    for name, blob in (("index", d.index_dict), ("connection", d.connection_dict), ("morph", d.morph_dict),
                       ("unk", d.unk_dict), ("char_category", d.char_category), ("invoke", d.invoke_list),
                       ("group", d.group_list)):
        a = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, bytes) else np.ascontiguousarray(blob, dtype=np.uint8)
        keep.append(a)
        setattr(b, name + "_p", a.ctypes.data if a.size else None)
        setattr(b, name + "_len", a.size)
    h = C.c_void_p()
    rc = _lib.lib().kgpu_dict_create(C.byref(b), 0, C.byref(h))
    if rc == 0:
        _lib.lib().kgpu_dict_destroy(h)
    return rc, _lib.lib().kgpu_last_error().decode()


def test_dict_validation_mirrors_reference_panics():
    p = fixture_dict_parts()
    good = Dict.from_parts(**p)
    rc, msg = _create(good)
    # valid dictionary: OK on a GPU box, NO_DEVICE here -- never a silent CPU fallback
    assert rc in (_lib.KGPU_OK, _lib.KGPU_ERR_NO_DEVICE), msg
    bad = dict(p); bad["morphs"] = p["morphs"][:2]  # keyword id 3 -> morphs[2] out of bounds (lattice.rs:182)
    assert _create(Dict.from_parts(**bad))[0] == _lib.KGPU_ERR_BAD_DICT
    bad = dict(p); bad["morphs"] = [[0, 0, 1], [1, 7, 1], [2, 2, 1]]  # right_id 7 outside the 3x3 matrix
    assert _create(Dict.from_parts(**bad))[0] == _lib.KGPU_ERR_BAD_DICT
    bad = dict(p); bad["unk_map"] = {1: (1, 3)}  # unk ids 1..3 but only 2 unk morphs (lattice.rs:195)
    assert _create(Dict.from_parts(**bad))[0] == _lib.KGPU_ERR_BAD_DICT
    bad = dict(p); bad["invoke_list"] = np.array([0, 1], dtype=np.uint8)  # category 2 indexes invoke_list[2]
    assert _create(Dict.from_parts(**bad))[0] == _lib.KGPU_ERR_BAD_DICT
    trunc = Dict.from_parts(**p); trunc.index_dict = trunc.index_dict[:-3]
    assert _create(trunc)[0] == _lib.KGPU_ERR_BAD_DICT


def test_no_cpu_fallback_without_device():
    if _lib.lib().kgpu_device_count() > 0:
        pytest.skip("GPU present")
    from kanpyo_amd import Tokenizer

    with pytest.raises(_lib.KgpuError) as e:
        Tokenizer(Dict.from_parts(**fixture_dict_parts()))
    assert e.value.code == _lib.KGPU_ERR_NO_DEVICE


def test_npz_cache_round_trip_and_stale_format(tmp_path, fixture_dict):
    """Dict.save_npz / load_npz: plain arrays only; a cache written by another revision is reported as stale (rebuild),
    not as an opaque numpy / json error."""
    from kanpyo_amd.dict import Dict

    p = tmp_path / "d.npz"
    fixture_dict.save_npz(p)
    back = Dict.load_npz(p)
    assert back.index_dict == fixture_dict.index_dict and back.char_class == fixture_dict.char_class
    np.savez_compressed(tmp_path / "old.npz", index_dict=np.zeros(4, dtype=np.uint8), char_class=np.array(["DEFAULT"]))
    with pytest.raises(ValueError, match="stale dictionary cache"):
        Dict.load_npz(tmp_path / "old.npz")
    with pytest.raises(FileNotFoundError):  # a missing file is not a stale cache
        Dict.load_npz(tmp_path / "absent.npz")
    (tmp_path / "junk.npz").write_bytes(b"not a zip archive")
    with pytest.raises(ValueError, match="stale dictionary cache"):
        Dict.load_npz(tmp_path / "junk.npz")

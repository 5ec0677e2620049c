"""A compiled C99 consumer of include/kanpyo_gpu.h (tests/c_abi/consumer.c) against the reference's fixture dictionary: the closest thing
to compiling the Rust shim of INTEGRATION.md this image allows.  It builds index.dict itself (kgpu_index_build), creates the dictionary,
tokenizes (one call for all sentences, then one call per sentence -- the reference's call shape, src/bin/kanpyo.rs:106-126), reads the 8-byte
record form and the routing counters through a device context, dumps a lattice, and prints every record; compared here with
tests/golden/fixture_tokens.json (hand-derived expected Vec<Token>) and with the oracle's lattice."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, fixture_dict_parts, load_golden

pytestmark = pytest.mark.gpu


def test_c_consumer_of_the_public_header(tmp_path):
    from kanpyo_amd import _lib
    from kanpyo_amd.dict import Dict, connection_blob, morphs_blob, unk_blob
    from kanpyo_amd.tokenizer import pack_sentences
    from oracle import pyref

    assert _lib.lib().kgpu_device_count() > 0
    exe = str(tmp_path / "consumer")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "consumer.c"), "-o", exe, "-L", libdir, "-lkanpyo_gpu", f"-Wl,-rpath,{libdir}"], check=True)
    p = fixture_dict_parts()
    d = tmp_path / "dict"
    d.mkdir()
    kws = [k.encode("utf-8") for k in p["sorted_keywords"]]
    (d / "keywords.bin").write_bytes(b"".join(kws))
    np.concatenate([[0], np.cumsum([len(k) for k in kws])]).astype("<u8").tofile(d / "keywords.off")
    (d / "connection.dict").write_bytes(connection_blob(p["conn_rows"], p["conn_cols"], p["conn_data"]))
    (d / "morph.dict").write_bytes(morphs_blob(p["morphs"]))
    (d / "unk.dict").write_bytes(unk_blob(p["unk_map"], p["unk_morphs"]))
    np.asarray(p["char_category"], dtype=np.uint8).tofile(d / "char_category.bin")
    np.asarray(p["invoke_list"], dtype=np.uint8).tofile(d / "invoke.bin")
    np.asarray(p["group_list"], dtype=np.uint8).tofile(d / "group.bin")
    cases = load_golden("fixture_tokens.json")["cases"]
    sents = [c["input"] for c in cases]
    utf8, offs = pack_sentences(sents)
    (d / "sentences.bin").write_bytes(utf8.tobytes())
    offs.astype("<u8").tofile(d / "sentences.off")

    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].split() == ["D", str(len(p["morphs"])), str(len(p["unk_morphs"])), str(p["conn_rows"]), str(p["conn_cols"])]
    # ---- tokens: golden (id, class, position, start, end, surface)
    it = iter(lines[1:])
    for i, case in enumerate(cases):
        tag, si, st, nt = next(it).split()
        assert (tag, int(si), int(st), int(nt)) == ("S", i, 0, len(case["tokens"])), (case["input"], tag, si, st, nt)
        raw = case["input"].encode("utf-8")
        for exp in case["tokens"]:
            f = next(it).split()
            assert f[0] == "T"
            tid, cls, pos, start, end, blen = map(int, f[1:])
            surface = "EOS" if cls == 0 else raw[pos:pos + blen].decode("utf-8")
            assert [tid, cls, pos, start, end, surface] == exp, case["input"]
    tag, batches, sentences = next(it).split()
    assert (tag, int(batches), int(sentences)) == ("R", 1, len(sents))
    tag, cus, waves = next(it).split()
    assert tag == "P" and int(cus) > 0 and int(waves) > 0
    # ---- lattice of sentence 0 against the naive Python restatement, node for node in insertion order (lattice.rs:105-110, 116-142)
    tag, n_nodes, n_pos = next(it).split()
    rd = Dict.from_parts(**p)
    pd = pyref.PyDict(rd.index_dict, rd.connection_dict, rd.morph_dict, rd.unk_dict, rd.char_category, rd.invoke_list, rd.group_list)
    e_nodes, _edges, e_dp, e_pre = pyref.lattice(pd, sents[0])
    nodes = [list(map(int, next(it).split()[1:])) for _ in range(int(n_nodes))]
    assert tag == "L" and int(n_pos) == len(sents[0]) + 2 and len(nodes) == len(e_nodes)
    for t, (got, (cls, nid, bpos, cpos, morph, bl, cl)) in enumerate(zip(nodes, e_nodes)):
        gid, gcls, gbpos, gcpos, gend, gbl, gl, gr, gc, gdp, gpre = got
        if t == 0:
            assert (gid, gcls, gpre) == (0, 0, -1)
            continue
        assert (gid, gcls, gbpos, gcpos, gbl) == (nid, cls, bpos, cpos, bl), t
        assert gend == (cpos + cl if cls else cpos), t
        if cls:
            assert (gl, gr, gc) == tuple(morph), t
        assert gdp == e_dp[t] and gpre == (-1 if e_pre[t] is None else e_pre[t]), t
    assert next(it, None) is None

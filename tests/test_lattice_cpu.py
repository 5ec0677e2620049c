"""the Graphviz rendering of a lattice (reference src/graphviz.rs:30-163), host side.  The DOT
writer is checked on the reference's fixture dictionary against a hand-derived rendering (the reference asserts none);
the lattice here comes from the naive Python restatement -- the device dump is compared with it in the gpu tests."""
import numpy as np

from conftest import fixture_dict_parts, load_golden
from kanpyo_amd.dict import Dict
from kanpyo_amd.lattice import Lattice, Node, graphviz
from kanpyo_amd.token import TokenClass
from oracle import pyref


def lattice_from_pyref(pd, text):
    """oracle/pyref.lattice -> kanpyo_amd.lattice.Lattice (shared with tests/test_gpu_parity.py)."""
    raw = text.encode("utf-8")
    nodes, edges, dp, pre = pyref.lattice(pd, text)
    out = []
    for i, (cls, nid, bpos, cpos, morph, bl, cl) in enumerate(nodes):
        out.append(Node(nid, TokenClass(cls), bpos, cpos, cpos + cl, morph[0], morph[1], morph[2], raw[bpos : bpos + bl].decode("utf-8"),
                        0 if dp[i] is None else dp[i], pre[i]))
    return Lattice(out, [list(e) for e in edges])


def _fixture():
    d = Dict.from_parts(**fixture_dict_parts())
    pd = pyref.PyDict(d.index_dict, d.connection_dict, d.morph_dict, d.unk_dict, d.char_category, d.invoke_list, d.group_list)
    return d, pd


def test_graphviz_known_answer_on_the_fixture():
    g = load_golden("fixture_graphviz.json")
    _, pd = _fixture()
    lat = lattice_from_pyref(pd, g["input"])
    conn = lambda r, l: pd.conn[pd.row * l + r]  # noqa: E731  ConnectionTable::get (connection.rs:12-14)
    known = lambda i: g["features"]["known"][str(i)]  # noqa: E731
    unk = lambda i: g["features"]["unknown"][str(i)]  # noqa: E731
    assert [n.surface for n in lat.viterbi()] == ["辞書", "形態素", ""]
    assert graphviz(lat, conn, known, unk, 48, False) == g["dot"]
    full = graphviz(lat, conn, known, unk, 48, True).split("\n")
    assert sum(1 for ln in full if " -- " in ln) == g["full_state_counts"]["edges"]
    assert sum(1 for ln in full if '[label="' in ln and " -- " not in ln) == g["full_state_counts"]["nodes"]  # (labels span lines)
    assert any("shape=diamond, color=red" in ln for ln in full)  # an Unknown node off the best path


def test_graphviz_degenerate_inputs():
    """"" (BOS and EOS compare equal as values: one visible node, no edge) and an unreachable EOS (empty best path)."""
    _, pd = _fixture()
    conn = lambda r, l: pd.conn[pd.row * l + r]  # noqa: E731
    f = lambda i: ["x"]  # noqa: E731
    dot = graphviz(lattice_from_pyref(pd, ""), conn, f, f)
    assert dot.count("[label=") == 1 and " -- " not in dot and '0 [label="BOS"' in dot
    lat = lattice_from_pyref(pd, "テ")  # category 0 has no unk entry: EOS has no predecessor
    assert lat.viterbi() == []
    dot = graphviz(lat, conn, f, f)
    # only EOS is reachable from EOS, and the reference labels visible id 0 "BOS" whatever it is (src/graphviz.rs:96-102)
    assert dot.count("[label=") == 1 and '0 [label="BOS"' in dot and "style=bold" not in dot

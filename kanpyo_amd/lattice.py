"""Lattice (reference src/lattice.rs:6-10) read back from the device, and the reference's Graphviz rendering of it
(src/graphviz.rs:9-163; CLI `kanpyo graphviz`, src/bin/kanpyo.rs:127-148) -- SURVEY.md 8(f) rank 4, the debugging aid
for a parity failure.  Host-side only: the lattice itself comes from kgpu_lattice_dump (include/kanpyo_gpu.h)."""
from __future__ import annotations

import ctypes as C
from collections import deque
from dataclasses import dataclass
from typing import List, Optional, Sequence

from . import _lib
from .token import TokenClass

INF = 1 << 30


@dataclass(frozen=True)
class Node:
    """src/lattice/node.rs:6-24 (Word / Node::Dummy) plus the Viterbi state of src/lattice.rs:118-141."""
    id: int
    class_: TokenClass
    byte_pos: int
    char_pos: int
    end_char: int
    left_id: int
    right_id: int
    cost: int
    surface: str
    dp: int
    pre: Optional[int]

    def key(self):
        """The derived `Ord` of `Node` (node.rs:6-24): variant first (Dummy < Known < Unknown), then the fields in order."""
        if self.class_ == TokenClass.Dummy:
            return (0, self.byte_pos, self.char_pos, (self.left_id, self.right_id, self.cost))
        return (int(self.class_), self.id, self.byte_pos, self.char_pos, (self.left_id, self.right_id, self.cost), self.surface.encode("utf-8"))


@dataclass
class Lattice:
    nodes: List[Node]
    edges: List[List[int]]  # edges[e] = indices of the nodes ENDING at char position e, ascending (lattice.rs:9)

    def viterbi(self) -> List[Node]:
        """The backtrace of Lattice::viterbi (lattice.rs:144-153) over the dumped predecessor links."""
        pos, path = len(self.nodes) - 1, []
        while self.nodes[pos].pre is not None:
            path.append(self.nodes[pos])
            pos = self.nodes[pos].pre
        return path[::-1]


def dump_lattice(tokenizer, text: str) -> Lattice:
    """Lattice::build + the forward pass of viterbi, on the device (kgpu_lattice_dump)."""
    raw = text.encode("utf-8")
    out = _lib.LatticeOut()
    buf = (C.c_uint8 * max(len(raw), 1)).from_buffer_copy(raw or b"\0")
    _lib.check(_lib.lib().kgpu_lattice_dump(tokenizer.handle, C.cast(buf, C.c_void_p), len(raw), C.byref(out)))
    try:
        nodes = []
        for i in range(out.n_nodes):
            n = out.nodes[i]
            cls = TokenClass(int(n.cls))
            nodes.append(Node(int(n.id), cls, int(n.byte_pos), int(n.char_pos), int(n.end_char), int(n.left_id), int(n.right_id), int(n.cost),
                              raw[n.byte_pos : n.byte_pos + n.byte_len].decode("utf-8"), int(n.dp), None if n.pre < 0 else int(n.pre)))
        eo = [int(out.edge_offsets[k]) for k in range(out.n_positions + 1)]
        edges = [[int(out.edge_nodes[k]) for k in range(eo[e], eo[e + 1])] for e in range(out.n_positions)]
        return Lattice(nodes, edges)
    finally:
        _lib.lib().kgpu_lattice_free(C.byref(out))


def graphviz(lat: Lattice, conn_get, known_features, unk_features, dpi: int = 48, full_state: bool = False) -> str:
    """Graphviz::graphviz (src/graphviz.rs:30-163) as a string.  conn_get(right_id, left_id) = ConnectionTable::get;
    known_features(id) / unk_features(id) = the feature names of a morph id (MorphFeatureTable)."""
    bests = {n.key() for n in lat.viterbi()}
    out = ["graph lattice {", f"dpi={dpi};",
           "graph [style=filled, splines=true, overlap=false, fontsize=30, rankdir=LR]",
           'edge [fontname=Helvetica, fontcolor=red, color="#606060"]',
           'node [shape=box, style=filled, fillcolor="#e8e8f0", fontname=Helvetica]']
    if not full_state:  # bfs from the last node over `edges[node.char_pos]`, unknown words only when on the best path (:10-28)
        visited, queue = {}, deque([lat.nodes[-1]])
        while queue:
            node = queue.popleft()
            if node.key() in visited:
                continue
            visited[node.key()] = node
            for i in lat.edges[node.char_pos]:
                m = lat.nodes[i]
                if m.key() not in visited and (m.class_ != TokenClass.Unknown or m.key() in bests):
                    queue.append(m)
        visible = [visited[k] for k in sorted(visited)]  # BTreeSet order
    else:
        visible = list(lat.nodes)
    dummy = TokenClass.Dummy
    for vid, n in enumerate(visible):
        if n.class_ == dummy:
            label = "BOS" if vid == 0 else "EOS"
        else:
            feats = known_features(n.id) if n.class_ == TokenClass.Known else unk_features(n.id)
            label = f"{n.surface}\n{'/'.join(f for f in feats if f != '*')}\n{n.cost}"
        color = {TokenClass.Known: "black", TokenClass.Unknown: "red", dummy: "blue"}[n.class_]
        if n.key() in bests or n.class_ == dummy:
            out.append(f'{vid} [label="{label}", shape=ellipse, color={color}, peripheries=2]')
        else:
            shape = {TokenClass.Known: "box", TokenClass.Unknown: "diamond", dummy: "ellipse"}[n.class_]
            out.append(f'{vid} [label="{label}", shape={shape}, color={color}]')
    vis_id = {}
    for vid, n in enumerate(visible):
        vis_id[n.key()] = vid  # BTreeMap from_iter: a later equal key replaces the earlier one
    for edge in lat.edges:
        for i in edge:
            node = lat.nodes[i]
            if node.key() not in vis_id:
                continue
            nid = vis_id[node.key()]
            for fi in lat.edges[node.char_pos]:
                frm = lat.nodes[fi]
                if frm.key() not in vis_id or vis_id[frm.key()] == nid:
                    continue
                label = str(conn_get(frm.right_id, node.left_id))
                ok1 = frm.key() in bests or frm.class_ == dummy
                ok2 = node.key() in bests or node.class_ == dummy
                if ok1 and ok2:
                    out.append(f'{vis_id[frm.key()]} -- {nid} [label="{label}", style=bold, color=blue, fontcolor=blue]')
                else:
                    out.append(f'{vis_id[frm.key()]} -- {nid} [label="{label}"]')
    out.append("}")
    return "\n".join(out) + "\n"


def graphviz_for(lat: Lattice, dictfile, dpi: int = 48, full_state: bool = False) -> str:
    """graphviz() wired to a loaded Kanpyo dictionary (kanpyo_amd.dictfile.DictFile)."""
    import struct

    import numpy as np

    c = dictfile.dict.connection_dict
    rows, _cols = struct.unpack_from("<QQ", c, 0)
    data = np.frombuffer(c, dtype="<i2", offset=16)
    return graphviz(lat, lambda r, l: int(data[rows * l + r]), dictfile.morph_feature_table.features, dictfile.unk_feature_table.features,
                    dpi, full_state)

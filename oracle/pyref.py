"""Pure-Python restatement of the reference hot path -- TEST INFRASTRUCTURE ONLY.

A second, independent and deliberately naive restatement (lists of lists,
exactly the data structures of the reference) used to cross-check the C oracle
on small cases.  Follows, line by line:
  src/lattice.rs:13-201, src/tokenizer.rs:16-45,
  kanpyo-dict/src/trie/da.rs:155-182, kanpyo-dict/src/index.rs:40-53,
  kanpyo-dict/src/connection.rs:12-14, kanpyo-dict/src/char_category_def.rs:33-38.
Parity vs the real reference binary is unpinned (it cannot be built here).
"""
from __future__ import annotations

import struct

import numpy as np

INF = 1 << 30
MAXIMUM_UNKNOWN_WORD_LENGTH = 1024


class Panic(Exception):
    """The reference would panic (index out of bounds) on this input."""


class PyDict:
    def __init__(self, index_dict, connection_dict, morph_dict, unk_dict, char_category, invoke_list, group_list):
        b = bytes(index_dict)
        (n,) = struct.unpack_from("<Q", b, 0)
        da = np.frombuffer(b, dtype="<i4", count=2 * n, offset=8).reshape(n, 2)
        self.base = da[:, 0].tolist()
        self.check = da[:, 1].tolist()
        o = 8 + 8 * n
        (m,) = struct.unpack_from("<Q", b, o)
        self.dup = {}
        for i in range(m):
            k, v = struct.unpack_from("<qQ", b, o + 8 + 16 * i)
            self.dup[k] = v
        c = bytes(connection_dict)
        self.row, self.col = struct.unpack_from("<QQ", c, 0)
        self.conn = np.frombuffer(c, dtype="<i2", count=self.row * self.col, offset=16).tolist()
        self.morphs = self._morphs(bytes(morph_dict), 0)[0]
        u = bytes(unk_dict)
        (k,) = struct.unpack_from("<Q", u, 0)
        self.unk = {}
        for i in range(k):
            cat, first, cnt = struct.unpack_from("<BqQ", u, 8 + 17 * i)
            self.unk[cat] = (first, cnt)
        self.unk_morphs = self._morphs(u, 8 + 17 * k)[0]
        self.cat = bytes(np.asarray(char_category, dtype=np.uint8).tobytes())
        self.invoke = [bool(x) for x in np.asarray(invoke_list, dtype=np.uint8)]
        self.group = [bool(x) for x in np.asarray(group_list, dtype=np.uint8)]

    @staticmethod
    def _morphs(b, o):
        (n,) = struct.unpack_from("<q", b, o)
        a = np.frombuffer(b, dtype="<i2", count=3 * n, offset=o + 8).reshape(n, 3)
        return [tuple(int(x) for x in r) for r in a], o + 8 + 6 * n

    def char_category(self, ch: int) -> int:
        return self.cat[ch] if ch < len(self.cat) else self.cat[0]

    def da_common_prefix(self, bs: bytes):
        p = 1
        out = []
        n = len(self.base)
        for i, ch in enumerate(bs):
            prev = p
            if prev >= n:
                raise Panic("da[prev]")
            p = self.base[prev] + ch
            if not (0 <= p < n) or self.check[p] != prev:
                break
            ahead = self.base[p] + 0
            if 0 <= ahead < n and self.check[ahead] == p and self.base[ahead] < 0:
                out.append((-self.base[ahead], i + 1))
        return out or None

    def search_common_prefix_of(self, bs: bytes):
        r = self.da_common_prefix(bs)
        if r is None:
            return None
        res = []
        for kid, ln in r:
            for i in range(self.dup.get(kid, 0) + 1):
                res.append((kid + i, ln))
        return res


def lattice(d: PyDict, text: str):
    """Lattice::build + the forward pass of Lattice::viterbi (src/lattice.rs:101-142) -> (nodes, edges, dp, pre) with
    nodes[i] = (cls, id, byte_pos, char_pos, morph, surface_bytes, surface_chars)."""
    inp = text.encode("utf-8")
    chars = list(text)
    C = len(chars)
    nodes = []  # (cls, id, byte_pos, char_pos, morph, surface_bytes, surface_chars)
    edges = [[] for _ in range(C + 2)]
    nodes.append((0, 0, 0, 0, (0, 0, 0), 0, 0))
    edges[0].append(0)
    byte_pos = 0
    for char_pos, ch in enumerate(chars):
        res = d.search_common_prefix_of(inp[byte_pos:])
        matched = res is not None
        if matched:
            for kid, bl in res:
                surface = inp[byte_pos : byte_pos + bl].decode("utf-8")  # panics if not boundary
                if not (0 <= kid - 1 < len(d.morphs)):
                    raise Panic("morphs")
                nodes.append((1, kid, byte_pos, char_pos, d.morphs[kid - 1], bl, len(surface)))
                edges[char_pos + len(surface)].append(len(nodes) - 1)
        cat = d.char_category(ord(ch))
        if (not matched) or d.invoke[cat]:
            is_group = d.group[cat] if cat < len(d.group) else False
            end_byte = byte_pos + len(ch.encode("utf-8"))
            ulen = 1
            if is_group:
                for nxt in chars[char_pos + 1 :]:
                    if d.char_category(ord(nxt)) != cat:
                        break
                    end_byte += len(nxt.encode("utf-8"))
                    ulen += 1
                    if ulen >= MAXIMUM_UNKNOWN_WORD_LENGTH:
                        break
            if cat in d.unk:
                first, cnt = d.unk[cat]
                for i in range(cnt):
                    uid = first + i
                    if not (0 <= uid - 1 < len(d.unk_morphs)):
                        raise Panic("unk morphs")
                    nodes.append((2, uid, byte_pos, char_pos, d.unk_morphs[uid - 1], end_byte - byte_pos, ulen))
                    edges[char_pos + ulen].append(len(nodes) - 1)
        byte_pos += len(ch.encode("utf-8"))
    nodes.append((0, 0, len(inp), C, (0, 0, 0), 0, 0))
    edges[C + 1].append(len(nodes) - 1)

    dp = [None] * len(nodes)
    pre = [None] * len(nodes)
    for char_pos in range(1, len(edges)):
        for i in edges[char_pos]:
            target = nodes[i]
            dp[i] = INF
            for j in edges[target[3]]:
                prev = nodes[j]
                prev_cost = dp[j] if dp[j] is not None else 0
                idx = d.row * (target[4][0] & 0xFFFFFFFFFFFFFFFF) + (prev[4][1] & 0xFFFFFFFFFFFFFFFF)
                if idx >= len(d.conn):
                    raise Panic("connection")
                total = min(prev_cost + target[4][2] + d.conn[idx], INF)
                if total < dp[i]:
                    dp[i] = total
                    pre[i] = j
    return nodes, edges, dp, pre


def tokenize(d: PyDict, text: str):
    """-> list of (id, cls, position, start, end, byte_len)."""
    nodes, edges, dp, pre = lattice(d, text)
    pos = len(nodes) - 1
    path = []
    while pre[pos] is not None:
        path.append(pos)
        pos = pre[pos]
    path.reverse()
    out = []
    for p in path:
        cls, nid, bpos, cpos, _m, bl, cl = nodes[p]
        if cls == 0:
            out.append((0, 0, bpos, cpos, cpos + 3, 0))
        else:
            out.append((nid, cls, bpos, cpos, cpos + cl, bl))
    return out

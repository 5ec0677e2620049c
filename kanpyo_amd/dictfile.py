"""Kanpyo `.dict` container: load / save.

The reference packs a dictionary as a zip (Deflate) of six blobs
(kanpyo-dict/src/dict.rs:51-116).  Four of them are the fixed-width little-endian
layouts the GPU boundary already speaks (kanpyo_amd/dict.py); `chardef.dict` and
`morph_feature.dict` (and the tail of `unk.dict`) are bincode 2
`config::standard()` encodings (char_category_def.rs:41-57,
morph_feature.rs:20-37): little endian, variable-length integers.

Host-side only (zipfile + bytes); nothing here is on the hot path.  Parity of the
two third-party container formats (zip 8, bincode 2) is UNPINNED: neither a
reference-built `.dict` nor the crates' sources are available here, so this
follows the published bincode 2 varint spec and round-trips against itself.
"""
from __future__ import annotations

import io
import struct
import zipfile
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

from .dict import Dict
from .token import Token, TokenClass

ENTRIES = ["morph.dict", "morph_feature.dict", "connection.dict", "index.dict", "chardef.dict", "unk.dict"]  # dict.rs:57-67


# ---------------------------------------------------------------- bincode 2 (standard config)

def enc_varint(v: int) -> bytes:
    if v < 0:
        raise ValueError("unsigned only")
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return b"\xfb" + struct.pack("<H", v)
    if v < 1 << 32:
        return b"\xfc" + struct.pack("<I", v)
    if v < 1 << 64:
        return b"\xfd" + struct.pack("<Q", v)
    return b"\xfe" + v.to_bytes(16, "little")


class _Reader:
    def __init__(self, b: bytes, at: int = 0):
        self.b, self.at = b, at

    def varint(self) -> int:
        t = self.b[self.at]
        self.at += 1
        if t < 251:
            return t
        n = {251: 2, 252: 4, 253: 8, 254: 16}.get(t)
        if n is None:
            raise ValueError("bincode: bad varint tag 255")
        v = int.from_bytes(self.b[self.at : self.at + n], "little")
        if len(self.b) < self.at + n:
            raise ValueError("bincode: truncated varint")
        self.at += n
        return v

    def take(self, n: int) -> bytes:
        if self.at + n > len(self.b):
            raise ValueError("bincode: truncated")
        out = self.b[self.at : self.at + n]
        self.at += n
        return out

    def string(self) -> str:
        return self.take(self.varint()).decode("utf-8")


@dataclass
class MorphFeatureTable:
    """morph_feature.rs:6-10: interned feature strings; name_list[0] == "" and ids start at 1."""

    morph_features: List[List[int]] = field(default_factory=list)
    name_list: List[str] = field(default_factory=lambda: [""])

    @classmethod
    def from_features(cls, rows: Sequence[Sequence[str]]) -> "MorphFeatureTable":
        """MorphFeatureTableBuilder (morph_feature.rs:39-100): ids in first-seen order from 1."""
        ids, names, out = {}, [""], []
        for row in rows:
            r = []
            for name in row:
                if name not in ids:
                    ids[name] = len(names)
                    names.append(name)
                r.append(ids[name])
            out.append(r)
        return cls(out, names)

    def encode(self) -> bytes:
        parts = [enc_varint(len(self.morph_features))]
        for row in self.morph_features:
            parts.append(enc_varint(len(row)))
            parts.extend(enc_varint(x) for x in row)
        parts.append(enc_varint(len(self.name_list)))
        for s in self.name_list:
            e = s.encode("utf-8")
            parts.append(enc_varint(len(e)) + e)
        return b"".join(parts)

    @classmethod
    def decode(cls, b: bytes, at: int = 0) -> Tuple["MorphFeatureTable", int]:
        r = _Reader(b, at)
        rows = [[r.varint() for _ in range(r.varint())] for _ in range(r.varint())]
        names = [r.string() for _ in range(r.varint())]
        return cls(rows, names), r.at

    def features(self, morph_id: int) -> List[str]:
        """`morph_features[id - 1]` mapped through name_list (src/bin/kanpyo.rs:178-183)."""
        return [self.name_list[i] for i in self.morph_features[morph_id - 1]]


def encode_chardef(char_class: Sequence[str], char_category, invoke_list, group_list) -> bytes:
    """CharCategoryDef bincode (char_category_def.rs:14-20,41-48)."""
    cat = np.ascontiguousarray(char_category, dtype=np.uint8).tobytes()
    parts = [enc_varint(len(char_class))]
    for s in char_class:
        e = s.encode("utf-8")
        parts.append(enc_varint(len(e)) + e)
    parts.append(enc_varint(len(cat)) + cat)
    for flags in (invoke_list, group_list):
        f = np.asarray(flags).astype(np.uint8).tobytes()
        parts.append(enc_varint(len(f)) + f)
    return b"".join(parts)


def decode_chardef(b: bytes):
    r = _Reader(b)
    char_class = [r.string() for _ in range(r.varint())]
    cat = np.frombuffer(r.take(r.varint()), dtype=np.uint8).copy()
    invoke = np.frombuffer(r.take(r.varint()), dtype=np.uint8).copy()
    group = np.frombuffer(r.take(r.varint()), dtype=np.uint8).copy()
    if (invoke > 1).any() or (group > 1).any():
        raise ValueError("chardef.dict: bool byte out of range")
    return char_class, cat, invoke, group


@dataclass
class DictFile:
    """A loaded `.dict`: the hot-path tables (`dict`) + the display tables."""

    dict: Dict
    morph_feature_table: MorphFeatureTable
    unk_feature_table: MorphFeatureTable


def _unk_prefix_len(unk: bytes) -> int:
    (k,) = struct.unpack_from("<Q", unk, 0)
    at = 8 + 17 * k
    (n,) = struct.unpack_from("<q", unk, at)
    return at + 8 + 6 * n


def load_dict(src) -> DictFile:
    """Dict::load (dict.rs:70-116) from a path, bytes or file object."""
    if isinstance(src, (bytes, bytearray)):
        src = io.BytesIO(src)
    with zipfile.ZipFile(src) as z:
        blobs = {name: z.read(name) for name in ENTRIES}
    char_class, cat, invoke, group = decode_chardef(blobs["chardef.dict"])
    unk = blobs["unk.dict"]
    cut = _unk_prefix_len(unk)
    d = Dict(blobs["index.dict"], blobs["connection.dict"], blobs["morph.dict"], unk[:cut], cat, invoke, group, char_class)
    return DictFile(d, MorphFeatureTable.decode(blobs["morph_feature.dict"])[0], MorphFeatureTable.decode(unk, cut)[0])


def save_dict(df: DictFile, dst) -> None:
    """Dict::build (dict.rs:51-69): zip, Deflate, the six entries in the reference's order."""
    d = df.dict
    blobs = {
        "morph.dict": d.morph_dict,
        "morph_feature.dict": df.morph_feature_table.encode(),
        "connection.dict": d.connection_dict,
        "index.dict": d.index_dict,
        "chardef.dict": encode_chardef(list(d.char_class), d.char_category, d.invoke_list, d.group_list),
        "unk.dict": d.unk_dict[: _unk_prefix_len(d.unk_dict)] + df.unk_feature_table.encode(),
    }
    with zipfile.ZipFile(dst, "w", compression=zipfile.ZIP_DEFLATED) as z:
        for name in ENTRIES:
            info = zipfile.ZipInfo(name)
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16  # unix_permissions(0o644), dict.rs:55
            z.writestr(info, blobs[name])


def format_tokens(tokens: Sequence[Token], df: DictFile) -> str:
    """The CLI's output lines, `surface\\tfeat,feat,...` (src/bin/kanpyo.rs:174-197)."""
    lines = []
    for t in tokens:
        if t.id != 0 and t.class_ == TokenClass.Known:
            feats = df.morph_feature_table.features(t.id)
        elif t.id != 0 and t.class_ == TokenClass.Unknown:
            feats = df.unk_feature_table.features(t.id)
        else:
            feats = []
        lines.append(f"{t.surface}\t{','.join(feats)}")
    return "\n".join(lines)

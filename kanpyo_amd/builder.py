"""MeCab-source dictionary builder -- host side, offline.

Mirrors the reference's `DictionaryBuilder::from_config`
(kanpyo-dict/src/builder.rs:46-116) and its parsers:
  lexicon CSVs   builder/record.rs:21-42   (surface,left,right,cost,features...)
  matrix.def     builder/matrix_def.rs:22-68
  char.def       builder/char_def.rs:22-99
  unk.def        builder/unk.rs:17-42  + unk_dict.rs:19-57
Token ids are defined by the sort order of the records (derived `Ord` of
`Record`, builder/record.rs:5-19), so that order is reproduced exactly; the
double array comes from `kgpu_index_build`, byte-identical to the reference's
packing.  The real mecab-ipadic sources are absent here, so this is exercised
on small hand-written MeCab-format inputs only: parity of a dictionary built here
from mecab-ipadic with the reference's own build is UNVERIFIED.
"""
from __future__ import annotations

import csv
import io
import os
import re
from typing import List, Sequence, Tuple

import numpy as np

from .dict import Dict
from .dictfile import DictFile, MorphFeatureTable

I16_MAX = 32767


class BuilderError(ValueError):
    """KanpyoError::{InvalidFormat, CostOutOfRange, CharCategoryNotFound, ...} or a reference panic."""


def _as_i16(v: int) -> int:
    """Rust `as i16` on an integer: wraps (builder.rs:65-69 casts usize/i64 with `as`)."""
    return ((v + 32768) & 0xFFFF) - 32768


def parse_matrix_def(text: str) -> Tuple[int, int, List[int]]:
    """builder/matrix_def.rs:22-68 -> (row, col, data) with data[c*row + r] = value."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    if not lines:
        raise BuilderError("matrix.def: missing 'row col' line")
    head = lines[0].split()
    if len(head) != 2:
        raise BuilderError(f"Invalid row and col: {lines[0]!r}")
    row, col = int(head[0]), int(head[1])
    data = [0] * (row * col)
    for line in lines[1:]:
        vals = [int(x) for x in line.split()]
        if len(vals) != 3:
            raise BuilderError(f"Invalid matrix value: {line!r}")
        r, c, v = vals
        if r < 0 or c < 0 or not (-32768 <= v <= 32767):
            raise BuilderError(f"matrix.def: value out of range: {line!r}")
        if r >= row or c >= col:
            raise BuilderError(f"Invalid matrix index: {line!r}")
        data[c * row + r] = v
    return row, col, data


_RE_CLASS = re.compile(r"^(\w+)\s+(\d+)\s+(\d+)\s+(\d+)")
_RE_POINT = re.compile(r"^(0x[0-9A-F]+)(?:\s+([^#\s]+))(?:\s+([^#\s]+))?")
_RE_RANGE = re.compile(r"^(0x[0-9A-F]+)..(0x[0-9A-F]+)(?:\s+([^#\s]+))(?:\s+([^#\s]+))?")


def parse_char_def(text: str):
    """builder/char_def.rs:35-99 -> (char_class, char_category[65536] u8, invoke, group).
    Only the FIRST category of a code-point line is used and the `length` column is ignored,
    exactly as the reference does."""
    char_class: List[str] = []
    cat = np.zeros(1 << 16, dtype=np.uint8)
    invoke: List[bool] = []
    group: List[bool] = []
    cc2id = {}
    for raw in text.split("\n"):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        m = _RE_CLASS.match(line)
        if m:
            invoke.append(m.group(2) == "1")
            group.append(m.group(3) == "1")
            cc2id[m.group(1)] = len(char_class) & 0xFF
            char_class.append(m.group(1))
            continue
        m = _RE_POINT.match(line)
        if m:
            ch = int(m.group(1)[2:], 16)
            if m.group(2) not in cc2id or ch >= cat.size:
                raise BuilderError(f"char.def: reference would panic on {line!r}")
            cat[ch] = cc2id[m.group(2)]
            continue
        m = _RE_RANGE.match(line)
        if m:
            lo, hi = int(m.group(1)[2:], 16), int(m.group(2)[2:], 16)
            if m.group(3) not in cc2id or hi >= cat.size:
                raise BuilderError(f"char.def: reference would panic on {line!r}")
            if lo <= hi:
                cat[lo : hi + 1] = cc2id[m.group(3)]
            continue
        raise BuilderError(f"Invalid char.def format: {line}")
    return char_class, cat, np.array(invoke, dtype=np.uint8), np.array(group, dtype=np.uint8)


def decode_euc_jp(data: bytes) -> str:
    """EUC-JP as encoding_rs::EUC_JP decodes it (the WHATWG Encoding Standard's decoder, which the reference
    uses: builder/record.rs:21-26): JIS X 0208 through the Windows-31J (CP932) table -- U+FF5E for 0xA1C1 where
    Python's own 'euc_jp' codec gives U+301C, likewise 0xA1C2 / 0xA1DD / 0xA1F1 / 0xA1F2 / 0xA2CC and NEC row 13 --
    half-width katakana behind 0x8E, JIS X 0212 behind 0x8F.  Surfaces decide the record sort order and with it
    every token id, so the table has to be the reference's.  Raises BuilderError("EncodingError") where the
    reference's `had_errors` is set."""
    out = []
    i, n = 0, len(data)
    try:
        while i < n:
            b = data[i]
            if b < 0x80:
                j = i + 1
                while j < n and data[j] < 0x80:
                    j += 1
                out.append(data[i:j].decode("ascii"))
                i = j
            elif b == 0x8E and i + 1 < n and 0xA1 <= data[i + 1] <= 0xDF:
                out.append(chr(0xFF61 - 0xA1 + data[i + 1]))
                i += 2
            elif b == 0x8F and i + 2 < n and 0xA1 <= data[i + 1] <= 0xFE and 0xA1 <= data[i + 2] <= 0xFE:
                out.append(data[i : i + 3].decode("euc_jp"))  # JIS X 0212
                i += 3
            elif 0xA1 <= b <= 0xFE and i + 1 < n and 0xA1 <= data[i + 1] <= 0xFE:
                ku, ten = b - 0xA0, data[i + 1] - 0xA0
                s1 = ((ku - 1) >> 1) + (0x81 if ku <= 62 else 0xC1)
                s2 = ten + 0x3F + (1 if ten >= 64 else 0) if ku & 1 else ten + 0x9E
                out.append(bytes((s1, s2)).decode("cp932"))
                i += 2
            else:
                raise UnicodeDecodeError("euc-jp", data, i, i + 1, "invalid EUC-JP sequence")
    except UnicodeDecodeError as e:
        raise BuilderError("EncodingError") from e
    return "".join(out)


def _rows(text: str):
    rows = [r for r in csv.reader(io.StringIO(text)) if r]
    for r in rows:  # csv::ReaderBuilder (flexible = false): every record has the first record's field count
        if len(r) != len(rows[0]):
            raise BuilderError(f"CSV error: record with {len(r)} fields, expected {len(rows[0])}: {r!r}")
    return rows


def parse_csv(text: str):
    """builder/record.rs:21-42 -> [(surface, left_id, right_id, cost, [features])]."""
    out = []
    for r in _rows(text):
        if len(r) < 4:
            raise BuilderError(f"lexicon row too short: {r!r}")
        out.append((r[0], int(r[1]), int(r[2]), int(r[3]), list(r[4:])))
        if out[-1][1] < 0 or out[-1][2] < 0:
            raise BuilderError(f"negative context id: {r!r}")  # usize parse error in the reference
    return out


parse_unk_def = parse_csv  # builder/unk.rs:31-42: (category, left, right, cost, features)


def _record_key(rec):
    """Derived Ord of Record / UnkDefRecord: String by bytes, then ids, cost, then Vec<String>."""
    return (rec[0].encode("utf-8"), rec[1], rec[2], rec[3], [f.encode("utf-8") for f in rec[4]])


def build(records: Sequence, matrix_def: str, char_def: str, unk_records: Sequence) -> DictFile:
    """DictionaryBuilder::from_config over already-decoded inputs (builder.rs:49-115)."""
    recs = sorted(records, key=_record_key)
    for r in recs:
        if r[3] > I16_MAX:
            raise BuilderError(f"Cost is too large: {r[3]}")  # builder.rs:60-62 panics
    keywords = [r[0] for r in recs]
    morphs = [[_as_i16(r[1]), _as_i16(r[2]), _as_i16(r[3])] for r in recs]
    features = MorphFeatureTable.from_features([r[4] for r in recs])
    row, col, data = parse_matrix_def(matrix_def)
    char_class, cat, invoke, group = parse_char_def(char_def)
    # UnkDict::build (unk_dict.rs:19-57)
    urecs = sorted(unk_records, key=_record_key)
    unk_morphs, unk_map = [], {}
    for i, r in enumerate(urecs):
        if r[3] > I16_MAX:
            raise BuilderError(f"CostOutOfRange({r[3]})")
        unk_morphs.append([_as_i16(r[1]), _as_i16(r[2]), _as_i16(r[3])])
        if r[0] not in char_class:
            raise BuilderError(f"CharCategoryNotFound({r[0]})")
        c = char_class.index(r[0]) & 0xFF
        first, cnt = unk_map.get(c, (i + 1, 0))
        unk_map[c] = (first, cnt + 1)
    d = Dict.from_parts(keywords, morphs if morphs else np.zeros((0, 3), dtype=np.int64), row, col, data, char_class, cat,
                        invoke, group, unk_map, unk_morphs if unk_morphs else np.zeros((0, 3), dtype=np.int64))
    return DictFile(d, features, MorphFeatureTable.from_features([r[4] for r in urecs]))


def build_from_dir(root: str, encoding: str = "euc_jp") -> DictFile:
    """`ipa_dict_builder --dict <root>` (bin/ipa_dict_builder.rs:38-59, builder/config.rs:18-27)."""
    def read(name, enc):
        with open(os.path.join(root, name), "rb") as f:
            raw = f.read()
        if enc.lower().replace("-", "_") == "euc_jp":
            return decode_euc_jp(raw)
        try:
            return raw.decode(enc)
        except UnicodeDecodeError as e:
            raise BuilderError("EncodingError") from e

    records = []
    for name in os.listdir(root):
        if name.endswith(".csv"):
            records += parse_csv(read(name, encoding))
    # matrix.def is read as plain text by the reference (matrix_def.rs:16-20)
    return build(records, read("matrix.def", "utf-8"), read("char.def", encoding), parse_unk_def(read("unk.def", encoding)))

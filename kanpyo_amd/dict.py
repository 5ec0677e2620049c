"""Dictionary container for the hot path (reference kanpyo-dict/src/dict.rs:20-30).

Holds the tables in the reference's own serialised form (DictReadWrite::write_dict,
SURVEY.md App. B), which is exactly what include/kanpyo_gpu.h:kgpu_dict_create
takes.  Constructors mirror the reference pieces a caller uses to assemble a
Dict by hand (reference src/tests.rs:8-108):
  IndexTable::build      kanpyo-dict/src/index.rs:16-38     -> index_table_build
  Morphs::from           kanpyo-dict/src/morph.rs:54-58      -> morphs_blob
  ConnectionTable::from  kanpyo-dict/src/connection.rs:17-25 -> connection_blob
  UnkDict{..}            kanpyo-dict/src/unk_dict.rs:12-16   -> unk_blob
The MorphFeatureTable (display strings) is not on the hot path and is not carried.
"""
from __future__ import annotations

import ctypes as C
import json
import struct
import zipfile
from dataclasses import dataclass, field
from typing import Iterable, Mapping, Sequence

import numpy as np

from . import _lib


def index_table_build(sorted_keywords: Iterable) -> bytes:
    """IndexTable::build + write_dict -> index.dict bytes (host only, no GPU needed)."""
    enc = [k.encode("utf-8") if isinstance(k, str) else bytes(k) for k in sorted_keywords]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.uint64, count=len(enc)))
    cat = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
    blob = C.c_void_p()
    blen = C.c_size_t(0)
    _lib.check(_lib.lib().kgpu_index_build(cat.ctypes.data, offs.ctypes.data, len(enc), C.byref(blob), C.byref(blen)))
    try:
        return C.string_at(blob, blen.value)
    finally:
        _lib.lib().kgpu_free(blob)


def morphs_blob(morphs) -> bytes:
    """Morphs::write_dict (morph.rs:61-72): i64 n; n x (i16 left, i16 right, i16 cost)."""
    a = np.ascontiguousarray(np.asarray(morphs, dtype=np.int64).reshape(-1, 3))
    if a.size and (a.min() < -32768 or a.max() > 32767):
        raise ValueError("morph field out of i16 range")
    return struct.pack("<q", a.shape[0]) + a.astype("<i2").tobytes()


def connection_blob(rows: int, cols: int, data) -> bytes:
    """ConnectionTable::write_dict (connection.rs:44-51); element (r, c) at c*rows + r."""
    a = np.ascontiguousarray(np.asarray(data, dtype=np.int64).reshape(-1))
    if a.size != rows * cols:
        raise ValueError("connection data must have rows*cols entries")
    return struct.pack("<QQ", rows, cols) + a.astype("<i2").tobytes()


def unk_blob(char_category_to_morph_id: Mapping[int, tuple], unk_morphs) -> bytes:
    """UnkDict::write_dict (unk_dict.rs:60-73) without the trailing feature table."""
    out = [struct.pack("<Q", len(char_category_to_morph_id))]
    for cat in sorted(char_category_to_morph_id):  # BTreeMap order
        first, count = char_category_to_morph_id[cat]
        out.append(struct.pack("<BqQ", cat, first, count))
    out.append(morphs_blob(unk_morphs))
    return b"".join(out)


NPZ_FORMAT = 2  # save_npz / load_npz: bumped whenever the set or encoding of the arrays changes


@dataclass
class Dict:
    index_dict: bytes
    connection_dict: bytes
    morph_dict: bytes
    unk_dict: bytes
    char_category: np.ndarray  # uint8, CharCategoryDef.char_category
    invoke_list: np.ndarray  # uint8 (bool), CharCategoryDef.invoke_list
    group_list: np.ndarray  # uint8 (bool), CharCategoryDef.group_list
    char_class: Sequence[str] = field(default_factory=list)

    def __post_init__(self):
        self.char_category = np.ascontiguousarray(self.char_category, dtype=np.uint8)
        self.invoke_list = np.ascontiguousarray(np.asarray(self.invoke_list).astype(np.uint8))
        self.group_list = np.ascontiguousarray(np.asarray(self.group_list).astype(np.uint8))

    @classmethod
    def from_parts(cls, sorted_keywords, morphs, conn_rows, conn_cols, conn_data, char_class, char_category,
                   invoke_list, group_list, unk_map, unk_morphs) -> "Dict":
        """Dict::new over freshly built tables (dict.rs:33-49, src/tests.rs:8-108)."""
        return cls(
            index_dict=index_table_build(sorted_keywords),
            connection_dict=connection_blob(conn_rows, conn_cols, conn_data),
            morph_dict=morphs_blob(morphs),
            unk_dict=unk_blob(unk_map, unk_morphs),
            char_category=char_category,
            invoke_list=invoke_list,
            group_list=group_list,
            char_class=list(char_class),
        )

    # -- small conveniences for sizing / reports
    @property
    def da_len(self) -> int:
        return struct.unpack_from("<Q", self.index_dict, 0)[0]

    @property
    def n_morphs(self) -> int:
        return struct.unpack_from("<q", self.morph_dict, 0)[0]

    @property
    def conn_shape(self):
        return struct.unpack_from("<QQ", self.connection_dict, 0)

    def save_npz(self, path):
        np.savez_compressed(
            path, index_dict=np.frombuffer(self.index_dict, dtype=np.uint8),
            connection_dict=np.frombuffer(self.connection_dict, dtype=np.uint8),
            morph_dict=np.frombuffer(self.morph_dict, dtype=np.uint8),
            unk_dict=np.frombuffer(self.unk_dict, dtype=np.uint8), char_category=self.char_category,
            invoke_list=self.invoke_list, group_list=self.group_list, format=np.int64(NPZ_FORMAT),
            char_class=np.frombuffer(json.dumps(list(self.char_class), ensure_ascii=False).encode("utf-8"), dtype=np.uint8),
        )

    @classmethod
    def load_npz(cls, path) -> "Dict":
        """Raises ValueError("stale dictionary cache ...") for a file written by another revision of save_npz (e.g. the
        pickled char_class of round 1): callers rebuild instead of failing inside numpy / json."""
        try:
            z = np.load(path, allow_pickle=False)  # plain arrays only: a dictionary cache must not be able to run code
            if "format" not in z.files or int(z["format"]) != NPZ_FORMAT:
                raise ValueError("format key missing or different")
            return cls(z["index_dict"].tobytes(), z["connection_dict"].tobytes(), z["morph_dict"].tobytes(),
                       z["unk_dict"].tobytes(), z["char_category"], z["invoke_list"], z["group_list"],
                       [str(x) for x in json.loads(z["char_class"].tobytes().decode("utf-8"))])
        except (ValueError, KeyError, UnicodeDecodeError, zipfile.BadZipFile) as e:  # (a missing or unreadable file is an OSError and stays one)
            raise ValueError(f"stale dictionary cache {path}: {e}; delete it and rebuild") from e

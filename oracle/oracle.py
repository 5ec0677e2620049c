"""ctypes wrapper around oracle/libkanpyo_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (kanpyo_amd/) never imports this module.
See oracle/kanpyo_oracle.h for what is restated and the parity status
("parity unpinned" at token level: the reference cannot be built here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TOKEN_DTYPE = np.dtype(
    [("id", "<i4"), ("cls", "<u4"), ("position", "<u4"), ("start", "<u4"), ("end", "<u4"), ("byte_len", "<u4")]
)

KORC_PANIC = -2
KORC_CAPACITY = -3
KORC_INVALID_UTF8 = -4


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sentences", "B", "C", "T", "N", "E", "K")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False, asan: bool = False) -> str:
    """Compile the oracle with gcc (building the checker is not using it)."""
    name = "libkanpyo_oracle_asan.so" if asan else "libkanpyo_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "kanpyo_oracle.c")
    hdr = os.path.join(_HERE, "kanpyo_oracle.h")
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in (src, hdr)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, name], check=True, capture_output=True)
    return path


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build(asan=bool(os.environ.get("KORC_ASAN"))))  # KORC_ASAN=1: the ASan/UBSan build (tests/test_oracle_sanitize.py)
        u8p = C.c_void_p
        L.korc_last_error.restype = C.c_char_p
        L.korc_dict_from_blobs.restype = C.c_void_p
        L.korc_dict_from_blobs.argtypes = [u8p, C.c_size_t] * 7
        L.korc_dict_free.argtypes = [C.c_void_p]
        L.korc_tokenize.restype = C.c_int64
        L.korc_tokenize.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.korc_tokenize_batch.restype = C.c_int
        L.korc_tokenize_batch.argtypes = [
            C.c_void_p, u8p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p,
        ]
        L.korc_tokenize_slots.restype = C.c_int
        L.korc_tokenize_slots.argtypes = [C.c_void_p, u8p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.korc_common_prefix.restype = C.c_int64
        L.korc_common_prefix.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.korc_da_search.restype = C.c_int64
        L.korc_da_search.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        L.korc_index_build.restype = C.c_void_p
        L.korc_index_build.argtypes = [u8p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.korc_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _buf(b):
    """bytes / ndarray -> (pointer, length, keepalive)."""
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray, memoryview)) else np.ascontiguousarray(b, dtype=np.uint8)
    return a.ctypes.data if a.size else None, a.size, a


def index_build(sorted_keywords) -> bytes:
    """IndexTable::build + write_dict (index.rs:16-38,75-84) -> index.dict blob."""
    enc = [k.encode("utf-8") if isinstance(k, str) else bytes(k) for k in sorted_keywords]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(e) for e in enc])
    cat = b"".join(enc)
    p, _, keep = _buf(cat if cat else b"\0")
    out_len = C.c_size_t(0)
    ptr = lib().korc_index_build(p, offs.ctypes.data, len(enc), C.byref(out_len))
    if not ptr:
        raise RuntimeError("oracle: reference would panic: " + lib().korc_last_error().decode())
    blob = C.string_at(ptr, out_len.value)
    lib().korc_free(ptr)
    return blob


def da_search(index_blob: bytes, key) -> int | None:
    k = key.encode("utf-8") if isinstance(key, str) else bytes(key)
    p, n, keep = _buf(index_blob)
    kp, kn, keep2 = _buf(k if k else b"")
    r = lib().korc_da_search(p, n, kp, kn)
    return int(r) if r != 0 else None


@dataclass
class OracleResult:
    tokens: np.ndarray  # TOKEN_DTYPE, dense
    offsets: np.ndarray  # uint64 [n+1]
    counters: dict


class OracleTokenizer:
    """CPU restatement of kanpyo::Tokenizer over serialised dictionary blobs."""

    def __init__(self, index_dict, connection_dict, morph_dict, unk_dict, char_category, invoke_list, group_list):
        parts = [index_dict, connection_dict, morph_dict, unk_dict, char_category, invoke_list, group_list]
        args, self._keep = [], []
        for b in parts:
            p, n, keep = _buf(b)
            args += [p, n]
            self._keep.append(keep)
        self._h = lib().korc_dict_from_blobs(*args)
        if not self._h:
            raise ValueError("oracle: " + lib().korc_last_error().decode())

    @classmethod
    def from_dict(cls, d):
        """d: any object with the blob attributes of kanpyo_amd.dict.Dict."""
        return cls(d.index_dict, d.connection_dict, d.morph_dict, d.unk_dict, d.char_category, d.invoke_list, d.group_list)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # `lib` is gone during interpreter shutdown
            try:
                lib().korc_dict_free(self._h)
            except Exception:
                pass
            self._h = None

    def common_prefix(self, text):
        s = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        ids = np.zeros(4096, dtype=np.int64)
        lens = np.zeros(4096, dtype=np.uint64)
        p, n, keep = _buf(s)
        m = lib().korc_common_prefix(self._h, p, n, ids.ctypes.data, lens.ctypes.data, 4096)
        if m < 0:
            raise RuntimeError("oracle panic")
        return None if m == 0 else [(int(ids[i]), int(lens[i])) for i in range(m)]

    def tokenize(self, text):
        """-> (list of (id, cls, position, start, end, byte_len), counters) ; raises on panic."""
        s = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        out = np.zeros(len(s) + 1, dtype=TOKEN_DTYPE)
        ctr = Counters()
        p, n, keep = _buf(s)
        k = lib().korc_tokenize(self._h, p, n, out.ctypes.data, out.size, C.byref(ctr))
        if k == KORC_INVALID_UTF8:
            raise UnicodeDecodeError("utf-8", s, 0, 1, "oracle: invalid UTF-8")
        if k < 0:
            raise RuntimeError("oracle: reference would panic: " + lib().korc_last_error().decode())
        return out[:k].copy(), ctr.as_dict()

    def tokenize_batch(self, utf8: np.ndarray, offsets: np.ndarray, nthreads: int = 1, out=None, copy: bool = True) -> OracleResult:
        """out=(tokens[TOKEN_DTYPE, >= bytes + n], tok_offsets[uint64, n + 1]): caller-owned, reusable result
        arrays (a timed caller keeps the allocation and its page faults out of the measurement);
        copy=False returns views of the result arrays instead of copies."""
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        cap = int(offsets[-1] - offsets[0]) + n
        if out is None:
            out_t, toff = np.zeros(cap, dtype=TOKEN_DTYPE), np.zeros(n + 1, dtype=np.uint64)
        else:
            out_t, toff = out
            if out_t.dtype != TOKEN_DTYPE or toff.dtype != np.uint64 or out_t.size < cap or toff.size < n + 1:
                raise ValueError("out=(tokens[TOKEN_DTYPE, >= bytes + n], tok_offsets[uint64, >= n + 1])")
        ctr = Counters()
        rc = lib().korc_tokenize_batch(
            self._h, utf8.ctypes.data if utf8.size else None, offsets.ctypes.data, n, out_t.ctypes.data, out_t.size,
            toff.ctypes.data, int(nthreads), C.byref(ctr),
        )
        if rc != 0:
            raise RuntimeError(f"oracle batch failed rc={rc}: " + lib().korc_last_error().decode())
        toks = out_t[: int(toff[n])]
        return OracleResult(toks.copy() if copy else toks, toff[: n + 1].copy() if copy and out is not None else toff[: n + 1], ctr.as_dict())

    def tokenize_slots(self, utf8: np.ndarray, offsets: np.ndarray, nthreads: int, reps: int = 1, out=None):
        """All-core timing form (korc_tokenize_slots): sentence i writes to out[offsets[i] - offsets[0] + i ..], its count to
        tok_count[i]; no allocation or merge copy inside the call.  -> (slots, tok_count, counters)."""
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        cap = int(offsets[-1] - offsets[0]) + n
        if out is None:
            out = (np.zeros(cap, dtype=TOKEN_DTYPE), np.zeros(max(n, 1), dtype=np.uint32))
        slots, cnt = out
        if slots.dtype != TOKEN_DTYPE or cnt.dtype != np.uint32 or slots.size < cap or cnt.size < n:
            raise ValueError("out=(slots[TOKEN_DTYPE, >= bytes + n], tok_count[uint32, >= n])")
        ctr = Counters()
        rc = lib().korc_tokenize_slots(self._h, utf8.ctypes.data if utf8.size else None, offsets.ctypes.data, n, slots.ctypes.data,
                                       cnt.ctypes.data, int(nthreads), int(reps), C.byref(ctr))
        if rc != 0:
            raise RuntimeError(f"oracle slots batch failed rc={rc}: " + lib().korc_last_error().decode())
        return slots, cnt[:n], ctr.as_dict()

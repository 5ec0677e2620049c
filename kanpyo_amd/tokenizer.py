"""Tokenizer (reference src/tokenizer.rs:7-45) over the HIP path.

    tokenizer = Tokenizer(dict)          # Tokenizer::new(dict)   src/tokenizer.rs:12-14
    tokens = tokenizer.tokenize("...")   # -> Vec<Token>          src/tokenizer.rs:16-45

plus the batched forms the GPU wants (many sentences per launch).  Everything
computes on the device through include/kanpyo_gpu.h; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from .dict import Dict
from .token import Token, TokenClass

# kgpu_token (include/kanpyo_gpu.h)
TOKEN_DTYPE = np.dtype(
    [("id", "<i4"), ("cls", "<u4"), ("position", "<u4"), ("start", "<u4"), ("end", "<u4"), ("byte_len", "<u4")]
)
TOKEN8_DTYPE = np.dtype([("id", "<i4"), ("packed", "<u4")])  # kgpu_token8: cls | chars << 2 | byte_len << 14


def pinned_empty(shape, dtype=np.uint8) -> np.ndarray:
    """np.empty in pinned host memory (kgpu_host_alloc); freed when the array (and its views) are collected."""
    import weakref

    dt = np.dtype(dtype)
    count = int(np.prod(shape)) if not np.isscalar(shape) else int(shape)
    nbytes = max(count * dt.itemsize, 1)
    L = _lib.lib()
    p = L.kgpu_host_alloc(nbytes)
    if not p:
        raise MemoryError(L.kgpu_last_error().decode("utf-8", "replace"))
    buf = (C.c_uint8 * nbytes).from_address(p)
    weakref.finalize(buf, L.kgpu_host_free, p)
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def pack_sentences(sentences: Sequence) -> tuple:
    """list of str/bytes -> (uint8 concatenation, uint64 offsets[n+1])."""
    enc = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in sentences]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.uint64, count=len(enc)))
    return np.frombuffer(b"".join(enc), dtype=np.uint8), offs


class Tokenizer:
    def __init__(self, dict: Dict, device: int = 0):
        self.dict = dict  # pub dict: Dict (src/tokenizer.rs:7-9)
        L = _lib.lib()
        b = _lib.DictBlobs()
        self._keep = []
        for name, blob in (
            ("index", dict.index_dict), ("connection", dict.connection_dict), ("morph", dict.morph_dict),
            ("unk", dict.unk_dict), ("char_category", dict.char_category), ("invoke", dict.invoke_list),
            ("group", dict.group_list),
        ):
            a = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray)) else np.ascontiguousarray(blob, dtype=np.uint8)
            self._keep.append(a)
            setattr(b, name + "_p", a.ctypes.data if a.size else None)
            setattr(b, name + "_len", a.size)
        h = C.c_void_p()
        _lib.check(L.kgpu_dict_create(C.byref(b), int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().kgpu_dict_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def info(self) -> dict:
        i = _lib.DictInfo()
        _lib.check(_lib.lib().kgpu_dict_get_info(self._h, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_ if n != "reserved"}

    # ---- packed batch: the form the C ABI speaks -------------------------------
    def tokenize_packed(self, utf8: np.ndarray, offsets: np.ndarray, token_capacity: int | None = None, pinned: bool = False,
                        out=None):
        """-> (tokens[TOKEN_DTYPE], tok_offsets[uint64 n+1], status[uint8 n]).
        pinned=True: the output arrays live in pinned host memory (kgpu_host_alloc), so the device-to-host
        copies of a large call run as DMA and overlap its kernels; pass inputs made with `pinned_empty` for
        the same effect on the way in.  out=(tokens, tok_offsets, status): caller-owned result arrays to reuse
        (a fresh 100 MB array costs more in page faults than the tokenization)."""
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        if n < 0:
            raise ValueError("offsets needs n+1 entries")
        total = int(offsets[-1] - offsets[0]) if n else 0
        cap = int(token_capacity) if token_capacity is not None else total // 2 + n + 64
        L = _lib.lib()
        while True:
            if out is not None:
                tokens, toff, status = out
                if tokens.dtype != TOKEN_DTYPE or toff.dtype != np.uint64 or status.dtype != np.uint8 or toff.size < n + 1 or status.size < n:
                    raise ValueError("out=(tokens[TOKEN_DTYPE], tok_offsets[uint64 >= n+1], status[uint8 >= n])")
                cap = tokens.size
                token_capacity = cap  # no silent reallocation of caller-owned arrays
            else:
                alloc = pinned_empty if pinned else np.empty
                tokens = alloc(cap, dtype=TOKEN_DTYPE)
                toff = alloc(n + 1, dtype=np.uint64)
                status = alloc(max(n, 1), dtype=np.uint8)
            status[: max(n, 1)] = 0
            got = C.c_uint64(0)
            rc = L.kgpu_tokenize_batch(
                self._h, utf8.ctypes.data if utf8.size else None, offsets.ctypes.data, n, tokens.ctypes.data, cap,
                toff.ctypes.data, status.ctypes.data, C.byref(got),
            )
            if rc == _lib.KGPU_ERR_CAPACITY and token_capacity is None:
                cap = int(got.value) + 64  # exact size reported by the device
                continue
            _lib.check(rc)
            return tokens[: int(got.value)], toff[: n + 1], status[:n]

    def routing(self, reset: bool = False) -> dict:
        """kgpu_dict_get_routing: the routing counters of the handle's pooled contexts (small_calls, combined_calls, ...)."""
        r = _lib.Routing()
        _lib.check(_lib.lib().kgpu_dict_get_routing(self._h, C.byref(r), C.sizeof(r), 1 if reset else 0))
        return {n: (list(getattr(r, n)) if n in ("deferred", "redone") else getattr(r, n)) for n, *_ in r._fields_}

    # ---- reference-shaped API --------------------------------------------------
    def tokenize_batch(self, sentences: Sequence[str]) -> List[List[Token]]:
        utf8, offs = pack_sentences(sentences)
        tokens, toff, status = self.tokenize_packed(utf8, offs)
        out = []
        for i, s in enumerate(sentences):
            if status[i] == _lib.KGPU_SENT_INVALID_UTF8:
                raise UnicodeDecodeError("utf-8", bytes(utf8[int(offs[i]) : int(offs[i + 1])]), 0, 1, "invalid UTF-8 sentence")
            raw = utf8[int(offs[i]) : int(offs[i + 1])].tobytes()
            row = []
            for t in tokens[int(toff[i]) : int(toff[i + 1])]:
                cls = TokenClass(int(t["cls"]))
                pos, bl = int(t["position"]), int(t["byte_len"])
                surface = "EOS" if cls == TokenClass.Dummy else raw[pos : pos + bl].decode("utf-8")
                row.append(Token(int(t["id"]), cls, pos, int(t["start"]), int(t["end"]), surface))
            out.append(row)
        return out

    def tokenize(self, input: str) -> List[Token]:
        """Tokenizer::tokenize (src/tokenizer.rs:16-45): one sentence == a batch of one."""
        return self.tokenize_batch([input])[0]


def tokenize_packed_multi(tokenizers: Sequence[Tokenizer], utf8: np.ndarray, offsets: np.ndarray, token_capacity: int | None = None, out=None, compact: bool = False):
    """kgpu_tokenize_batch_multi: sentence i -> tokenizers[i mod G] (one Tokenizer per device; the same one may appear more than once), results in
    the caller's original order -- byte for byte what Tokenizer.tokenize_packed gives on one device.
    -> (tokens[TOKEN_DTYPE], tok_offsets[uint64 n+1], status[uint8 n]).
    compact=True: kgpu_tokenize_batch_multi_compact -> (tokens8[TOKEN8_DTYPE], first[uint32 n x 2], tok_offsets, status); kanpyo_amd.device.expand_tokens
    (kgpu_expand_tokens) restores the 24-byte records where they are consumed."""
    utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    if n < 0:
        raise ValueError("offsets needs n+1 entries")
    total = int(offsets[-1] - offsets[0]) if n else 0
    cap = int(token_capacity) if token_capacity is not None else total // 2 + n + 64
    L = _lib.lib()
    handles = (C.c_void_p * len(tokenizers))(*[t.handle for t in tokenizers])
    first = np.zeros((max(n, 1), 2), dtype=np.uint32) if compact else None
    while True:
        if out is not None:
            tokens, toff, status = out
            cap = tokens.size
            token_capacity = cap
        else:
            tokens = np.empty(cap, dtype=TOKEN8_DTYPE if compact else TOKEN_DTYPE)
            toff = np.empty(n + 1, dtype=np.uint64)
            status = np.empty(max(n, 1), dtype=np.uint8)
        status[: max(n, 1)] = 0
        got = C.c_uint64(0)
        u = utf8.ctypes.data if utf8.size else None
        if compact:
            rc = L.kgpu_tokenize_batch_multi_compact(handles, len(tokenizers), u, offsets.ctypes.data, n, tokens.ctypes.data, cap, first.ctypes.data,
                                                     toff.ctypes.data, status.ctypes.data, C.byref(got))
        else:
            rc = L.kgpu_tokenize_batch_multi(handles, len(tokenizers), u, offsets.ctypes.data, n, tokens.ctypes.data, cap, toff.ctypes.data, status.ctypes.data, C.byref(got))
        if rc == _lib.KGPU_ERR_CAPACITY and token_capacity is None:
            cap = int(got.value) + 64
            continue
        _lib.check(rc)
        if compact:
            return tokens[: int(got.value)], first[:n], toff[: n + 1], status[:n]
        return tokens[: int(got.value)], toff[: n + 1], status[:n]


def concurrent_callers(tok: Tokenizer, utf8: np.ndarray, offsets: np.ndarray, threads: int, calls_per_thread: int, n_pattern=(1,), expect=None) -> dict:
    """Measurement / test helper (kgpu_debug_concurrent_callers, not part of the public header): `threads` native host threads call
    kgpu_tokenize_batch in a loop -- thread t with n_pattern[t % len] sentences per call -- walking round the corpus.  expect=(tokens, offsets)
    of the whole corpus (e.g. the oracle's): every call's records are compared, `mismatching_calls` counts the ones that differ."""
    utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    L = _lib.lib()
    f = L.kgpu_debug_concurrent_callers
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    pat = np.ascontiguousarray(n_pattern, dtype=np.int32)
    stats = np.zeros(8, dtype=np.float64)
    et = eo = None
    if expect is not None:
        et = np.ascontiguousarray(expect[0]); eo = np.ascontiguousarray(expect[1], dtype=np.uint64)
        assert et.dtype == TOKEN_DTYPE
    _lib.check(f(tok.handle, utf8.ctypes.data, offsets.ctypes.data, offsets.size - 1, int(threads), int(calls_per_thread), pat.ctypes.data, pat.size,
                 et.ctypes.data if et is not None else None, eo.ctypes.data if eo is not None else None, stats.ctypes.data))
    return {"wall_s": float(stats[0]), "p50_us": float(stats[1]), "p99_us": float(stats[2]), "mean_us": float(stats[3]), "mismatching_calls": int(stats[4]),
            "calls": int(stats[5]), "sentences": int(stats[6]), "sentences_per_s": float(stats[6] / stats[0]) if stats[0] > 0 else 0.0,
            "threads": int(threads), "n_pattern": [int(x) for x in pat], "caller_cpu_s": float(stats[7])}


TOKEN8_DTYPE = np.dtype([("id", "<i4"), ("packed", "<u4")])  # kgpu_token8


def merge_shards(shards, cnt: int, slice_sentences: int = 2048, reps: int = 1, token_capacity: int | None = None, want_tokens: bool = True, compact: bool = False):
    """Measurement / test helper (kgpu_debug_merge_shards[_compact], not part of the public header; needs NO device): the host-side merge of
    kgpu_tokenize_batch_multi[_compact] over one super-chunk of `cnt` sentences.  shards[g] = (rec[TOKEN8_DTYPE], first[uint32 m x 2], toff[uint64 m + 1],
    status[uint8 m]) as shard g's compaction kernel leaves them; sentence j of the super-chunk is shard j mod G's local sentence j // G.
    -> (rc, tokens, tok_offsets, status, n_tokens, seconds for all `reps` repetitions); compact: tokens = (tokens8, first[cnt x 2])."""
    G = len(shards)
    L = _lib.lib()
    vp = C.c_void_p
    keep = [[np.ascontiguousarray(a, dtype=dt) for a, dt in zip(sh, (TOKEN8_DTYPE, np.uint32, np.uint64, np.uint8))] for sh in shards]
    arr = lambda k: (vp * G)(*[sh[k].ctypes.data for sh in keep])
    total = sum(int(sh[2][-1]) for sh in keep)
    cap = total if token_capacity is None else int(token_capacity)
    tokens = np.zeros(max(cap, 1), dtype=TOKEN8_DTYPE if compact else TOKEN_DTYPE)
    toff = np.zeros(cnt + 1, dtype=np.uint64)
    status = np.full(max(cnt, 1), 255, dtype=np.uint8)
    n_tok, secs = C.c_uint64(0), C.c_double(0)
    if compact:
        f = L.kgpu_debug_merge_shards_compact
        f.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_uint64, C.c_int, vp, vp, C.c_uint64, vp, vp,
                      C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        f.restype = C.c_int
        first = np.full((max(cnt, 1), 2), 0xABABABAB, dtype=np.uint32)
        rc = f(G, cnt, arr(0), arr(1), arr(2), arr(3), int(slice_sentences), int(reps), tokens.ctypes.data if want_tokens else None, first.ctypes.data, cap,
               toff.ctypes.data, status.ctypes.data, C.byref(n_tok), C.byref(secs))
        return rc, (tokens[: min(cap, total)], first[:cnt]), toff, status[:cnt], int(n_tok.value), float(secs.value)
    f = L.kgpu_debug_merge_shards
    f.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp,
                  C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    f.restype = C.c_int
    rc = f(G, cnt, arr(0), arr(1), arr(2), arr(3), int(slice_sentences), int(reps), tokens.ctypes.data if want_tokens else None, cap, toff.ctypes.data,
           status.ctypes.data, C.byref(n_tok), C.byref(secs))
    return rc, tokens[: min(cap, total)], toff, status[:cnt], int(n_tok.value), float(secs.value)


def merge_bench(G: int = 8, sentences_per_shard: int = 8192, tokens_per_sentence: int = 32, reps: int = 20, compact: bool = False) -> dict:
    """The rate of that merge alone on this host's CPUs (bench.py's `multi_merge` entry): G synthetic shard blocks of a super-chunk, every sentence
    `tokens_per_sentence` records.  Per sentence the merge reads 8 t + 17 bytes and writes 24 t + 9 (t tokens): the 24-byte expansion is a
    memory-bandwidth job, so the rate is quoted beside a plain copy of the same number of bytes by the same worker threads' count of NumPy threads."""
    import time

    cnt = G * sentences_per_shard
    rng = np.random.default_rng(5)
    shards = []
    for g in range(G):
        m = sentences_per_shard
        toff = (np.arange(m + 1, dtype=np.uint64) * np.uint64(tokens_per_sentence))
        nt = int(toff[-1])
        rec = np.zeros(nt, dtype=TOKEN8_DTYPE)
        rec["id"] = rng.integers(1, 390000, size=nt)
        rec["packed"] = 1 | (2 << 2) | (6 << 14)
        shards.append((rec, np.zeros((m, 2), dtype=np.uint32), toff, np.zeros(m, dtype=np.uint8)))
    merge_shards(shards, cnt, reps=2, compact=compact)
    rc, _, _, _, n_tok, secs = merge_shards(shards, cnt, reps=reps, compact=compact)
    _lib.check(rc)
    moved = reps * ((n_tok * 16 + cnt * 34) if compact else (n_tok * 32 + cnt * 26))
    a = np.ones(n_tok * 24 // 8, dtype=np.uint64); b = np.empty_like(a)
    b[:] = a
    t0 = time.perf_counter()
    for _ in range(5):
        b[:] = a
    copy_gbs = 5 * a.nbytes * 2 / (time.perf_counter() - t0) / 1e9
    return {"sentences_per_s": reps * cnt / secs, "G": G, "sentences_per_super_chunk": cnt, "tokens_per_sentence": n_tok / cnt,
            "bytes_moved_GB_per_s": moved / secs / 1e9, "one_thread_copy_GB_per_s": copy_gbs, "record_bytes": 8 if compact else 24,
            "what": "kgpu_tokenize_batch_multi" + ("_compact" if compact else "") + "'s merge alone (no device): G shards' 8-byte records -> the caller's order as " + ("8" if compact else "24") + "-byte records + global offsets + "
                    "status bytes; slice totals from the shards' offset tables on the calling thread, one worker-pool task per 2048 sentences walks the G "
                    "cursors (no division per sentence); the calling thread's own share is O(slices x G)"}

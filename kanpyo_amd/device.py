"""Device-resident entry points (include/kanpyo_gpu.h: kgpu_ctx_*, kgpu_tokenize_device).

Inputs already in HBM, outputs left in HBM: what bench.py times and what the
multi-GPU gather (kanpyo_amd/dist.py) consumes.  Pointers cross as plain
integers (e.g. torch.Tensor.data_ptr()); torch itself is not imported here.
"""
from __future__ import annotations

import ctypes as C

from . import _lib

PROFILE_OFF, PROFILE_EVENTS, PROFILE_WORK, PROFILE_SAMPLED, PROFILE_NO_T = 0, 1, 2, 4, 8
STAGE_ALL, STAGE_LATTICE, STAGE_GATHER, STAGE_VITERBI = 0, 5, 6, 7  # kgpu_ctx_set_ablation


class DeviceContext:
    """One HIP stream + scratch arena; not thread-safe, make one per thread / per queue slot."""

    def __init__(self, tokenizer, stream_ptr: int | None = None):
        self._tok = tokenizer  # keeps the dictionary alive
        h = C.c_void_p()
        _lib.check(_lib.lib().kgpu_ctx_create(tokenizer.handle, C.c_void_p(stream_ptr) if stream_ptr else None, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().kgpu_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tokenize(self, d_utf8: int, d_offsets: int, n: int, total_bytes: int, d_tokens: int, token_capacity: int,
                 d_tok_offsets: int, d_status: int):
        """Enqueue one batch (asynchronous)."""
        _lib.check(_lib.lib().kgpu_tokenize_device(
            self._h, C.c_void_p(d_utf8), C.c_void_p(d_offsets), n, total_bytes, C.c_void_p(d_tokens), token_capacity,
            C.c_void_p(d_tok_offsets), C.c_void_p(d_status)))

    def tokenize_compact(self, d_utf8: int, d_offsets: int, n: int, total_bytes: int, d_tokens8: int, token_capacity: int,
                         d_first: int, d_tok_offsets: int, d_status: int):
        """Same batch, results as 8-byte kgpu_token8 records + the first token's (position, start) per sentence: a third of
        the volume for whatever moves them next (PCIe, the xGMI gather); expand_tokens() restores the 24-byte records."""
        _lib.check(_lib.lib().kgpu_tokenize_device_compact(
            self._h, C.c_void_p(d_utf8), C.c_void_p(d_offsets), n, total_bytes, C.c_void_p(d_tokens8), token_capacity,
            C.c_void_p(d_first), C.c_void_p(d_tok_offsets), C.c_void_p(d_status)))

    def sync(self) -> int:
        """Wait for the enqueued batch; returns its dense token count."""
        n = C.c_uint64(0)
        _lib.check(_lib.lib().kgpu_ctx_sync(self._h, C.byref(n)))
        return int(n.value)

    def set_profiling(self, mode: int):
        _lib.check(_lib.lib().kgpu_ctx_set_profiling(self._h, int(mode)))

    def profile(self, reset: bool = True) -> dict:
        """Event timings (kgpu_profile) and routing counters (kgpu_routing) in one dict."""
        p, r = _lib.Profile(), _lib.Routing()
        _lib.check(_lib.lib().kgpu_ctx_get_profile(self._h, C.byref(p), int(reset)))
        _lib.check(_lib.lib().kgpu_ctx_get_routing(self._h, C.byref(r), C.sizeof(r), int(reset)))
        return {"launches": int(p.launches), "tokenize_ms": float(p.tokenize_ms), "aux_ms": float(p.aux_ms),
                "batches": int(r.batches), "sentences": int(r.sentences), "deferred": [int(x) for x in r.deferred],
                "redone": [int(x) for x in r.redone], "long_launches": int(r.long_launches),
                "arena_regrows": int(r.arena_regrows), "first_ms": float(r.first_ms),
                "small_calls": int(r.small_calls), "small_fallbacks": int(r.small_fallbacks), "window_reruns": int(r.window_reruns), "tail_reruns": int(r.tail_reruns)}

    def plan(self) -> dict:
        """The launch plan (kgpu_plan_info): LDS bytes per workgroup and resident workgroups per CU of both kernels."""
        p = _lib.PlanInfo()
        _lib.check(_lib.lib().kgpu_ctx_get_plan(self._h, C.byref(p), C.sizeof(p)))
        return {n: int(getattr(p, n)) for n, _ in p._fields_ if n != "reserved"}

    def set_ablation(self, stop_after_stage: int):
        """Measurement only: following batches stop after the given stage (STAGE_*), zero tokens; 0 = off."""
        _lib.check(_lib.lib().kgpu_ctx_set_ablation(self._h, int(stop_after_stage)))

    def work(self, reset: bool = True) -> dict:
        w = _lib.Work()
        _lib.check(_lib.lib().kgpu_ctx_get_work(self._h, C.byref(w), int(reset)))
        return {n: int(getattr(w, n)) for n, _ in w._fields_}

    def phase_cycles(self, reset: bool = True) -> dict:
        """Per-phase shader-clock cycles of the LDS kernel, summed over sentences (PROFILE_WORK runs)."""
        arr = (C.c_uint64 * 10)()
        _lib.check(_lib.lib().kgpu_ctx_get_phase_cycles(self._h, C.byref(arr), int(reset)))
        names = ["load", "decode", "walk", "scan", "emit", "gather", "sweep", "backtrace_tokens", "sentences", "spare"]
        return {n: int(arr[i]) for i, n in enumerate(names)}


def expand_tokens(tokens8, tok_offsets, first):
    """kgpu_expand_tokens: 8-byte records (uint32 [T, 2] or uint64 [T]) + token offsets [n + 1] + first [n, 2] -> TOKEN_DTYPE [T]
    (host side, numpy arrays)."""
    import numpy as np

    from .tokenizer import TOKEN_DTYPE

    t8 = np.ascontiguousarray(tokens8).view(np.uint32).reshape(-1, 2)
    toff = np.ascontiguousarray(tok_offsets, dtype=np.uint64)
    fs = np.ascontiguousarray(first, dtype=np.uint32).reshape(-1)
    n = toff.size - 1
    out = np.empty(int(toff[n] - toff[0]), dtype=TOKEN_DTYPE)
    if out.size:
        _lib.lib().kgpu_expand_tokens(t8[int(toff[0]):].ctypes.data, toff.ctypes.data, fs.ctypes.data, n, out.ctypes.data)
    return out

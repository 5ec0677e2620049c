"""ctypes binding of libkanpyo_gpu.so (include/kanpyo_gpu.h).  Fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KGPU_LIB: another build of the same library (kernel experiments, tools/ only); the default is the in-tree build
LIB_PATH = os.environ.get("KGPU_LIB") or os.path.join(_HERE, "libkanpyo_gpu.so")

KGPU_OK = 0
KGPU_ERR_INVALID_ARG = 1
KGPU_ERR_BAD_DICT = 2
KGPU_ERR_HIP = 3
KGPU_ERR_CAPACITY = 4
KGPU_ERR_NO_DEVICE = 5
KGPU_ERR_INTERNAL = 6
KGPU_SENT_OK = 0
KGPU_SENT_INVALID_UTF8 = 1
KGPU_SENT_TRUNCATED = 3

# every symbol include/kanpyo_gpu.h declares
SYMBOLS = [
    "kgpu_last_error", "kgpu_device_count", "kgpu_dict_create", "kgpu_dict_destroy", "kgpu_dict_get_info",
    "kgpu_tokenize_batch", "kgpu_ctx_create", "kgpu_ctx_destroy", "kgpu_tokenize_device", "kgpu_tokenize_device_compact", "kgpu_expand_tokens", "kgpu_ctx_sync",
    "kgpu_ctx_set_profiling", "kgpu_ctx_set_ablation", "kgpu_ctx_get_profile", "kgpu_ctx_get_routing", "kgpu_ctx_get_plan", "kgpu_ctx_get_work", "kgpu_ctx_get_phase_cycles", "kgpu_index_build", "kgpu_free",
    "kgpu_host_alloc", "kgpu_host_free", "kgpu_lattice_dump", "kgpu_lattice_free",
    "kgpu_dict_get_routing", "kgpu_tokenize_batch_multi", "kgpu_tokenize_batch_multi_compact", "kgpu_multi_create", "kgpu_multi_destroy", "kgpu_multi_tokenize_device", "kgpu_multi_sync",
]


def kernel_source_hash() -> str:
    """sha256[:16] over every source and header the library is built from (the Makefile's SRCS and HDRS: kernels, device helpers,
    shared structs, the runtime's launch planning).  Profile-derived files under profiles/ record it, bench.py compares: a counter
    file measured on other code says so (`stale`)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("kgpu_pool.hip", "kgpu_kernels.hip", "kgpu_window.hip", "kgpu_device.h", "kgpu_internal.h", "kgpu_chartrie.cpp",
                 "kgpu_api.cpp", "kgpu_multi.cpp", "kgpu_runtime.h", "kgpu_index_build.cpp", "../../include/kanpyo_gpu.h"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


class KgpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kgpu error {code}: {msg}")
        self.code = code


class DictBlobs(C.Structure):
    _fields_ = [
        (n, t)
        for name in ("index", "connection", "morph", "unk", "char_category", "invoke", "group")
        for n, t in ((name + "_p", C.c_void_p), (name + "_len", C.c_size_t))
    ]


class DictInfo(C.Structure):
    _fields_ = [
        ("da_len", C.c_uint64), ("n_morphs", C.c_uint64), ("n_unk_morphs", C.c_uint64), ("conn_rows", C.c_uint64),
        ("conn_cols", C.c_uint64), ("device_bytes", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32),
    ]


class Profile(C.Structure):  # kgpu_profile: 24 bytes, frozen
    _fields_ = [("launches", C.c_uint64), ("tokenize_ms", C.c_double), ("aux_ms", C.c_double)]


class Routing(C.Structure):  # kgpu_routing: read with its size, fields are only ever appended
    _fields_ = [("batches", C.c_uint64), ("sentences", C.c_uint64), ("deferred", C.c_uint64 * 4), ("redone", C.c_uint64 * 4),
                ("long_launches", C.c_uint64), ("arena_regrows", C.c_uint64), ("first_ms", C.c_double),
                ("small_calls", C.c_uint64), ("small_fallbacks", C.c_uint64), ("window_reruns", C.c_uint64), ("tail_reruns", C.c_uint64),
                ("combined_calls", C.c_uint64), ("combined_launches", C.c_uint64)]


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("compute_units", "pool_lds_bytes", "pool_wavefronts", "pool_workgroups_per_cu", "pool_max_pages",
                                          "long_lds_bytes", "long_workgroups_per_cu", "long_workgroups", "window_lds_bytes", "window_workgroups_per_cu", "window_workgroups",
                                          "streams", "long_streams", "window_first_bytes")] + [("reserved", C.c_uint32 * 2)]


class LatticeNode(C.Structure):
    _fields_ = [("id", C.c_int32), ("cls", C.c_uint32), ("byte_pos", C.c_uint32), ("char_pos", C.c_uint32), ("end_char", C.c_uint32),
                ("byte_len", C.c_uint32), ("left_id", C.c_int16), ("right_id", C.c_int16), ("cost", C.c_int16), ("reserved", C.c_int16),
                ("dp", C.c_int32), ("pre", C.c_int32)]


class LatticeOut(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_positions", C.c_uint64), ("nodes", C.POINTER(LatticeNode)),
                ("edge_offsets", C.POINTER(C.c_uint32)), ("edge_nodes", C.POINTER(C.c_uint32))]


class Token(C.Structure):  # kgpu_token
    _fields_ = [("id", C.c_int32), ("cls", C.c_uint32), ("position", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32), ("byte_len", C.c_uint32)]


class Token8(C.Structure):  # kgpu_token8
    _fields_ = [("id", C.c_int32), ("packed", C.c_uint32)]


class Work(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sentences", "B", "C", "T", "N", "E", "K")]


_lib = None


def lib():
    """Load the in-tree HIP extension; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C kanpyo_amd/csrc). "
                "kanpyo_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.kgpu_last_error.restype = C.c_char_p
        L.kgpu_device_count.restype = C.c_int
        L.kgpu_dict_create.argtypes = [C.POINTER(DictBlobs), C.c_int, C.POINTER(vp)]
        L.kgpu_dict_destroy.argtypes = [vp]
        L.kgpu_dict_destroy.restype = None
        L.kgpu_dict_get_info.argtypes = [vp, C.POINTER(DictInfo)]
        L.kgpu_tokenize_batch.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.POINTER(C.c_uint64)]
        L.kgpu_ctx_create.argtypes = [vp, vp, C.POINTER(vp)]
        L.kgpu_ctx_destroy.argtypes = [vp]
        L.kgpu_ctx_destroy.restype = None
        L.kgpu_tokenize_device.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp]
        L.kgpu_tokenize_device_compact.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp, vp]
        L.kgpu_expand_tokens.argtypes = [vp, vp, vp, C.c_uint64, vp]
        L.kgpu_expand_tokens.restype = None
        L.kgpu_ctx_sync.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.kgpu_ctx_set_profiling.argtypes = [vp, C.c_int]
        L.kgpu_ctx_set_ablation.argtypes = [vp, C.c_int]
        L.kgpu_ctx_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int]
        L.kgpu_ctx_get_routing.argtypes = [vp, C.POINTER(Routing), C.c_size_t, C.c_int]
        L.kgpu_dict_get_routing.argtypes = [vp, C.POINTER(Routing), C.c_size_t, C.c_int]
        L.kgpu_ctx_get_plan.argtypes = [vp, C.POINTER(PlanInfo), C.c_size_t]
        L.kgpu_ctx_get_work.argtypes = [vp, C.POINTER(Work), C.c_int]
        L.kgpu_ctx_get_phase_cycles.argtypes = [vp, C.POINTER(C.c_uint64 * 10), C.c_int]
        L.kgpu_index_build.argtypes = [vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.kgpu_free.argtypes = [vp]
        L.kgpu_free.restype = None
        L.kgpu_host_alloc.argtypes = [C.c_uint64]
        L.kgpu_host_alloc.restype = vp
        L.kgpu_host_free.argtypes = [vp]
        L.kgpu_host_free.restype = None
        L.kgpu_lattice_dump.argtypes = [vp, vp, C.c_uint64, C.POINTER(LatticeOut)]
        L.kgpu_lattice_free.argtypes = [C.POINTER(LatticeOut)]
        L.kgpu_lattice_free.restype = None
        L.kgpu_tokenize_batch_multi.argtypes = [C.POINTER(vp), C.c_int, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.POINTER(C.c_uint64)]
        L.kgpu_tokenize_batch_multi_compact.argtypes = [C.POINTER(vp), C.c_int, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, C.POINTER(C.c_uint64)]
        L.kgpu_multi_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]
        L.kgpu_multi_destroy.argtypes = [vp]
        L.kgpu_multi_destroy.restype = None
        L.kgpu_multi_tokenize_device.argtypes = [vp, C.c_int] + [vp] * 9
        L.kgpu_multi_sync.argtypes = [vp, C.c_int, vp]
        _lib = L
    return _lib


def check(rc: int):
    if rc != KGPU_OK:
        raise KgpuError(rc, lib().kgpu_last_error().decode("utf-8", "replace"))

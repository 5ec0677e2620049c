"""The C boundary checked against the header itself: a C99 program that includes only include/kanpyo_gpu.h prints sizeof / offsetof of
every struct, and the ctypes mirror (kanpyo_amd/_lib.py) and the numpy token dtype must agree field by field.  The consumer program of
tests/c_abi/ must compile as strict C99 against the header and link against libkanpyo_gpu.so (running it needs a GPU: test_c_abi_gpu.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

from conftest import ROOT
from kanpyo_amd import _lib

HERE = os.path.join(ROOT, "tests", "c_abi")
INC = os.path.join(ROOT, "include")


def _layout(tmp_path):
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", INC, os.path.join(HERE, "layout.c"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    lay = {}
    for line in out.splitlines():
        st, f, off, size = line.split()
        lay.setdefault(st, {})[f] = (int(off), int(size))
    return lay


# header struct -> (ctypes mirror, {header field: mirror field} where the names differ)
_BLOB_NAMES = {h: m for name in ("index", "connection", "morph", "unk") for h, m in ((f"{name}_dict", f"{name}_p"), (f"{name}_len", f"{name}_len"))}
_BLOB_NAMES.update({"char_category": "char_category_p", "char_category_len": "char_category_len", "invoke_list": "invoke_p", "invoke_len": "invoke_len",
                    "group_list": "group_p", "group_len": "group_len"})
MIRRORS = {
    "kgpu_token": (_lib.Token, {}),
    "kgpu_token8": (_lib.Token8, {}),
    "kgpu_dict_blobs": (_lib.DictBlobs, _BLOB_NAMES),
    "kgpu_dict_info": (_lib.DictInfo, {}),
    "kgpu_profile": (_lib.Profile, {}),
    "kgpu_routing": (_lib.Routing, {}),
    "kgpu_plan_info": (_lib.PlanInfo, {}),
    "kgpu_work": (_lib.Work, {}),
    "kgpu_lattice_node": (_lib.LatticeNode, {}),
    "kgpu_lattice": (_lib.LatticeOut, {}),
}


def test_struct_layouts_of_the_header_match_the_ctypes_mirror(tmp_path):
    lay = _layout(tmp_path)
    assert set(lay) == set(MIRRORS), set(lay) ^ set(MIRRORS)
    for st, (mirror, names) in MIRRORS.items():
        fields = dict(lay[st])
        assert fields.pop("-") == (0, C.sizeof(mirror)), (st, "sizeof")
        mirror_fields = {n for n, *_ in mirror._fields_}
        assert {names.get(f, f) for f in fields} == mirror_fields, (st, {names.get(f, f) for f in fields} ^ mirror_fields)
        for f, (off, size) in fields.items():
            m = getattr(mirror, names.get(f, f))
            assert (m.offset, m.size) == (off, size), (st, f, (m.offset, m.size), (off, size))


def test_token_dtype_matches_the_header(tmp_path):
    from kanpyo_amd.tokenizer import TOKEN_DTYPE

    lay = _layout(tmp_path)["kgpu_token"]
    assert TOKEN_DTYPE.itemsize == lay["-"][1]
    for f, (off, size) in lay.items():
        if f != "-":
            assert TOKEN_DTYPE.fields[f][1] == off and TOKEN_DTYPE.fields[f][0].itemsize == size, f
    from kanpyo_amd.tokenizer import TOKEN8_DTYPE  # the 8-byte record of the gather / the host pipeline

    assert TOKEN8_DTYPE.itemsize == C.sizeof(_lib.Token8) and TOKEN8_DTYPE.fields["packed"][1] == _lib.Token8.packed.offset


def test_header_is_strict_c99_and_the_consumer_links(tmp_path):
    exe = str(tmp_path / "consumer")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", INC, os.path.join(HERE, "consumer.c"), "-o", exe,
                    "-L", os.path.dirname(_lib.LIB_PATH), "-lkanpyo_gpu", f"-Wl,-rpath,{os.path.dirname(_lib.LIB_PATH)}"], check=True)
    # every kgpu_* symbol the consumer references resolves against the library (it links); running it needs a device
    syms = subprocess.run(["nm", "-u", exe], check=True, capture_output=True, text=True).stdout
    used = {w for line in syms.splitlines() for w in line.split() if w.startswith("kgpu_")}
    assert {"kgpu_dict_create", "kgpu_tokenize_batch", "kgpu_ctx_get_routing", "kgpu_lattice_dump", "kgpu_index_build", "kgpu_expand_tokens"} <= used

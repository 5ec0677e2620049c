#!/usr/bin/env python
"""Do consecutive launches of ONE stream overlap when they carry no barrier bit (hipExtAnyOrderLaunch)?  The pool kernel over the same
4096-sentence batch, reps launches back to back on one stream, nothing in between (kgpu_debug_pool_repeat; measurement only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kanpyo_amd import Tokenizer, synth, _lib
from kanpyo_amd.device import DeviceContext
from kanpyo_amd.tokenizer import pack_sentences
sd = synth.build_dict(); sents = synth.make_corpus(sd, 4096, 1, "cfg2")
tok = Tokenizer(sd.dict); dev = torch.device("cuda", 0)
u, o = pack_sentences(sents)
du, do = torch.from_numpy(u.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
L = _lib.lib(); L.kgpu_debug_pool_repeat.restype = C.c_double
L.kgpu_debug_pool_repeat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
ctx = DeviceContext(tok)
for any_order in (0, 1, 0, 1):
    L.kgpu_debug_pool_repeat(ctx._h, du.data_ptr(), do.data_ptr(), 4096, int(o[-1]), 20, any_order)
    ms = L.kgpu_debug_pool_repeat(ctx._h, du.data_ptr(), do.data_ptr(), 4096, int(o[-1]), 200, any_order)
    print(f"one stream, 200 launches of 4096 sentences, any_order={any_order}: {ms:.2f} ms = {200 * 4096 / ms / 1e3:.1f} M sentences/s", flush=True)

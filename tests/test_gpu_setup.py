"""Process set-up the library does for itself: concurrent launches need hardware queues (GPU_MAX_HW_QUEUES, read when the HIP runtime initialises).
Loaded before the first HIP call with the variable unset, libkanpyo_gpu.so sets it to 8 and runs on four streams; with the variable set below 5 -- or loaded
too late -- it runs on three and says so (kgpu_plan_info.streams, a warning in kgpu_last_error after kgpu_dict_create).  Each case in a fresh process."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PROBE = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch                      # (imported first, as in bench.py and the tests: import alone makes no HIP call)
from kanpyo_amd import Tokenizer, _lib, synth
if os.environ.get("PROBE_LATE"):
    torch.cuda.is_available()     # a HIP call BEFORE the library is loaded: too late for it to ask for queues
L = _lib.lib()
sd = synth.build_dict(6000, seed=3)
tok = Tokenizer(sd.dict)
warn = L.kgpu_last_error().decode()
from kanpyo_amd.device import DeviceContext
plan = DeviceContext(tok).plan()
import ctypes
getenv = ctypes.CDLL(None).getenv          # the C environment (os.environ is Python's start-up snapshot: it does not see the library's setenv)
getenv.restype = ctypes.c_char_p
env = getenv(b"GPU_MAX_HW_QUEUES")
print(json.dumps({"streams": plan["streams"], "warning": warn, "env": env.decode() if env else None}))
""" % ROOT


def _probe(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "KGPU_STREAMS")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", PROBE], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads(r.stdout.strip().split("\n")[-1])


def test_library_asks_for_its_hardware_queues_itself():
    got = _probe({})
    assert got == {"streams": 4, "warning": "", "env": "16"}, got


def test_three_streams_are_reported_loudly():
    got = _probe({"GPU_MAX_HW_QUEUES": "4"})
    assert got["streams"] == 3 and "warning" in got["warning"] and "GPU_MAX_HW_QUEUES" in got["warning"], got
    late = _probe({"PROBE_LATE": "1"})
    assert late["streams"] == 3 and late["env"] is None and "warning" in late["warning"], late

import json
import os
import sys

import numpy as np
import pytest

try:  # torch ships its own libamdhip64: load it BEFORE libkanpyo_gpu.so so the process has one HIP runtime
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

os.environ.setdefault("KGPU_TEST_HOOKS_REREAD", "1")  # the runtime's test-only hooks (KGPU_NO_SMALL_CALLS, KGPU_HOST_CHUNK_*) are re-read per call
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


def fixture_dict_parts():
    """The reference's create_test_dict() (src/tests.rs:8-108) as plain data."""
    g = load_golden("fixture_dict.json")
    cat = np.zeros(g["char_category_len"], dtype=np.uint8)
    for r in g["char_category_ranges"]:
        cat[ord(r["from"]) : ord(r["to"]) + 1] = r["cat"]
    unk_map = {int(k): tuple(v) for k, v in g["char_category_to_morph_id"].items()}
    return dict(
        sorted_keywords=g["sorted_keywords"], morphs=g["morphs"], conn_rows=g["connection"]["row"],
        conn_cols=g["connection"]["col"], conn_data=g["connection"]["data"], char_class=g["char_class"],
        char_category=cat, invoke_list=np.array(g["invoke_list"], dtype=np.uint8),
        group_list=np.array(g["group_list"], dtype=np.uint8), unk_map=unk_map, unk_morphs=g["unk_morphs"],
    )


@pytest.fixture(scope="session")
def fixture_dict():
    from kanpyo_amd.dict import Dict

    return Dict.from_parts(**fixture_dict_parts())


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle

    oracle.build()
    return oracle


def has_gpu():
    try:
        from kanpyo_amd import _lib

        return _lib.lib().kgpu_device_count() > 0
    except Exception:
        return False

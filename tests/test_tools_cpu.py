"""The profile post-processing that bench.py's `roofline.traffic` / `instruction` fields come from (tools/pmc_summary.py,
tools/make_traffic_json.py): per-launch figures must be taken over FULL-batch launches of the dominant kernel, not over
the mix of ragged batches and single-launch small calls a profiled run also holds."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POOL = "void kgpu::k_tokenize_pool<false, false>(kgpu::PoolArgs)"
PROF = "void kgpu::k_tokenize_pool<true, false>(kgpu::PoolArgs)"  # the profiling instantiation (work counters): never the one the figures are taken from
COLS = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value"]


def _write_pass(root, k, rows):
    d = os.path.join(root, f"pass{k}", "x")
    os.makedirs(d)
    with open(os.path.join(d, "1_counter_collection.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=COLS)
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(COLS, r)))


def test_traffic_is_per_full_batch_launch(tmp_path):
    root = str(tmp_path / "pmc")
    # two full batches (grid 1024 x 256) and three small launches; FETCH_SIZE in KB, 10 per sentence
    sizes = [(1, 262144, 4096), (2, 262144, 4096), (3, 256, 1), (4, 4096, 64), (5, 108544, 1696)]
    prof = [(0, 262144, 4096)]  # a full batch through the profiling instantiation comes FIRST in the run (as in bench.py): other figures, to be ignored
    _write_pass(root, 1, [(i, g, k, c, v * f) for k, rows, f in ((PROF, prof, 2), (POOL, sizes, 1)) for i, g, n in rows
                          for c, v in (("SQ_WAVES", n // f), ("SQ_INSTS_VALU", 3000 * n), ("SQ_INSTS_SALU", 2000 * n), ("SQ_INSTS_LDS", 600 * n), ("SQ_INSTS_VMEM_RD", 100 * n), ("SQ_WAVE_CYCLES", 25000 * n))])
    _write_pass(root, 2, [(i, g, k, "FETCH_SIZE", 10 * n * f) for k, rows, f in ((PROF, prof, 3), (POOL, sizes, 1)) for i, g, n in rows])
    _write_pass(root, 3, [(i, g, k, "WRITE_SIZE", 1 * n * f) for k, rows, f in ((PROF, prof, 3), (POOL, sizes, 1)) for i, g, n in rows])
    summ, out = str(tmp_path / "s.json"), str(tmp_path / "pmc_traffic.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), root, "--json", summ], check=True, capture_output=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), summ, out, "t"], check=True, capture_output=True)
    t = json.load(open(out))
    assert t["batches"] == 2 and t["fetch_size_kb"] == 40960 and t["write_size_kb"] == 4096
    assert t["hbm_bytes_per_launch"] == (2 * 40960 + 4096) * 1024
    ins = json.load(open(str(tmp_path / "pmc_instructions.json")))
    assert ins["valu_per_sentence"] == 3000 and ins["salu_per_sentence"] == 2000 and ins["wave_cycles_per_sentence"] == 100000
    s = json.load(open(summ))
    assert s[POOL.split("(")[0]]["FETCH_SIZE"]["dispatches"] == 5  # the mix is still reported, under the plain name
    # both files name the kernel sources they were measured on; bench.py flags them when the tree has moved on
    sys.path.insert(0, ROOT)
    from kanpyo_amd._lib import kernel_source_hash

    assert t["kernel_src_sha16"] == ins["kernel_src_sha16"] == kernel_source_hash() and len(kernel_source_hash()) == 16

"""bench.py's result line (round 5's 24 KB line was not parsed by the driver: BENCH_r05.json parsed = null): compact_line() over a canned full report of
the size a real run produces stays within 6 KB, round-trips as JSON, and carries the keys the contract and the review ask for."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402

PI = 3.141592653589793


def canned_full(multi=False):
    """A report shaped like bench.py's `full` dict, every float with 16 digits and every string longer than it is in a real run."""
    extra_line = {"workload": "w" * 300, "sentences": 1000000, "value": PI * 1e7, "unit": "sentences/s", "Mchar_per_s": PI * 1e3, "input_MiB_per_s": PI * 1e3,
                  "roofline_at_job_rate": {"achieved": PI * 500, "peak": 8000.0, "unit": "GB/s", "frac": PI / 10}, "slot_occupancy": PI / 4,
                  "slot_occupancy_what": "x" * 400, "first_batch_bit_exact_vs_oracle": True, "launch_plan": {f"k{i}": i for i in range(14)},
                  "work_per_sentence": {k: PI * 100 for k in "BCTNEK"}, "routing": {"deferred": [0] * 4, "redone": [0] * 4}, "cpu_1thread": {"value": PI * 1e4}}
    stages = {n: {"bytes_per_step": 1e9, "ms_per_step": PI / 10, "achieved": PI * 1e3, "frac": PI / 10} for n in ("A_lattice", "B_viterbi", "C_emit")}
    full = {
        "metric": "sentences/sec", "value": PI * 3e7, "unit": "sentences/s", "n_gpus": 8 if multi else 1, "steps": 20, "warmup": 5, "ms_per_step": PI / 3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1] (cfg 2): " + "y" * 300, "batch": 4096, "sentences_per_step": 100000, "batches_per_step_per_gpu": 25,
                   "batches_in_flight": 8, "streams": 4, "long_streams": 8, "GPU_MAX_HW_QUEUES": "16", "residency": "z" * 60, "sharding": "s" * 100},
        "input_MiB_per_s": PI * 1e4, "work_per_sentence": {k: PI * 100 for k in "BCTNEK"}, "routing": {"deferred": [0] * 4}, "launch_plan": {f"k{i}": i for i in range(14)},
        "roofline": {"bound": "hbm", "achieved": PI * 200, "peak": 8000.0, "unit": "GB/s", "frac": PI / 40, "traffic": PI * 1e7, "traffic_stale": False,
                     "traffic_source": "t" * 300, "kernel": bench.KERNEL, "algorithmic_bytes_per_sentence": PI * 7e3, "algorithmic_bytes_per_launch": PI * 3e7,
                     "stage_bytes_per_launch": {"A_lattice": 1.0, "B_viterbi": 2.0, "C_emit": 3.0}, "avg_kernel_ms": PI / 25, "launches_timed": 128,
                     "avg_launch_chain_ms": PI / 24, "aux_kernels_avg_ms": PI / 200, "launches_in_flight": 8, "achieved_at_job_rate": PI * 700, "frac_at_job_rate": PI / 12,
                     "kernel_alone_ms": PI / 40, "frac_alone": PI / 23, "peak_measured_read": PI * 2e3, "frac_of_measured_read": PI / 30, "stages": stages,
                     "stage_A_ms": PI / 10, "stage_A_frac": PI / 15, "stage_B_ms": PI / 10, "stage_B_frac": PI / 9.5, "stage_C_ms": PI / 50, "stage_C_frac": PI / 24,
                     "instruction": {"valu_per_sentence": 3370.0, "what": "i" * 300}, "valu_issue_frac": PI / 12, "valu_issue_frac_3cyc": PI / 8, "insts_per_sentence": PI * 2e3},
        "sentences_total": 2000000,
        "pcie_inclusive": {"value": PI * 1e7, "call_latency": {f"n{n}": {"median_us": PI * 17, "p10_us": PI * 16, "sentences_per_s_at_median": PI * 1e4} for n in (1, 64, 4096)},
                           "concurrent_callers": {**{f"threads{n}": {"sentences_per_s": PI * 1e5, "p99_us": PI * 100} for n in (1, 16, 64, 128)},
                                                  "threads128_sustained": {"sentences_per_s": PI * 1e5}, "host_cpus": 16},
                           "large_call_pageable": PI * 2e7, "large_call_pinned": PI * 2.5e7},
        "multi_merge": {"sentences_per_s": PI * 6e7, "compact": {"sentences_per_s": PI * 2e8}, "host_cpus": 16},
        "value_end_to_end": {"value": PI * 2.5e7, "unit": "sentences/s"},
        "cpu_baseline": {"value": PI * 3e4, "unit": "sentences/s", "cores": 1, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "passes": 5,
                         "passes_spread": PI / 100, "pinned": True, "sample": "c" * 200, "all_cores": {"value": PI * 4e5, "cores": 16, "passes": 12}, "gpu_batch0_bit_exact": True},
        "speedup_vs_cpu_1thread": PI * 400,
        "extra": {k: dict(extra_line, stages=stages) for k in ("dense", "cfg5_q8", "cfg5_one_batch", "cfg3_b65536", "cfg3_b4096", "one_ctx_b4096", "one_ctx_b16384", "real_dict")},
    }
    if multi:
        full["gather"] = {"chunks": 20, "tokens": 10**9, "sentences": 2000000, "complete": True, "reassembled_step_equals_one_gpu": True, "chunk_steps": 1,
                          "record_bytes": 8, "root_ingest_GB_per_s": PI * 10, "records": "r" * 300}
        full["per_rank"] = [{"rank": r, "sentences": 250000, "seconds": PI / 100, "sentences_per_s": PI * 1e7} for r in range(8)]
        full["one_gpu_leg"] = {"value": PI * 3e7, "what": "o" * 200}
        full["speedup_vs_1gpu"] = PI * 2
    return full


def test_line_fits_and_round_trips():
    for multi in (False, True):
        full = canned_full(multi)
        assert len(json.dumps(full)) > 12000  # the full report is the size that broke round 5's record
        text = bench.fit_line(bench.compact_line(full))
        assert "\n" not in text and len(text) <= 6144, len(text)
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                  "cpu_baseline", "value_end_to_end", "value_dense", "stage_B_frac_dense", "extra_summary", "full_report"):
            assert k in line, k
        assert line["config"]["workload"].startswith("BASELINE configs[1]") and "model" not in line["config"]
        r = line["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "stage_A_frac", "stage_B_frac", "stage_C_frac", "stage_B_ms", "frac_at_job_rate", "frac_alone",
                  "avg_kernel_ms", "kernel", "valu_issue_frac", "peak_measured_read", "algorithmic_bytes_per_launch"):
            assert k in r, k
        assert len(r) <= 24 and all(not isinstance(v, (dict, list)) for v in r.values())
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        for k in ("value", "unit", "cores", "kind", "cpu_model", "sample", "passes_spread"):
            assert k in line["cpu_baseline"], k
        assert abs(line["value"] - full["value"]) / full["value"] < 1e-4  # five significant digits
        for k in ("cfg3_b65536", "cfg3_b4096", "cfg5_q8", "cfg5_one_batch", "dense", "one_ctx_b4096"):
            e = line["extra_summary"][k]
            assert set(e) >= {"value", "frac_at_job_rate", "slot_occupancy", "bit_exact"} and len(e) <= 6
        assert set(line["extra_summary"]["call_latency_us"]) == {"n1", "n64", "n4096"}
        if multi:
            assert line["gather"]["complete"] is True and len(line["per_rank_sentences_per_s"]) == 8 and "speedup_vs_1gpu" in line


def test_line_degrades_instead_of_failing():
    """A report that would not fit (a hundred extra legs) still yields a parseable line with the contract's keys."""
    full = canned_full()
    full["extra"].update({f"leg{i}": full["extra"]["dense"] for i in range(100)})
    line = json.loads(bench.fit_line(bench.compact_line(full)))
    assert "extra_summary" not in line and line["roofline"]["stage_B_frac"] and line["cpu_baseline"]["value"]


def test_bench_py_is_short():
    """The review's bound: bench.py = the headline and the line (<= 500 lines); the job in bench_engine.py, the other legs in bench_extras.py."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert sum(1 for _ in open(os.path.join(root, "bench.py"))) <= 500

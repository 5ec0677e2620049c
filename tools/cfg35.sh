export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
timeout 150 python tools/bench_cfg.py cfg3 200000 2>&1 | tail -1
timeout 150 python tools/bench_cfg.py cfg5 8000 1000 2>&1 | tail -1

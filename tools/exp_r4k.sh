#!/bin/bash
cd /root/repo
out=gpurun_out/r4k; mkdir -p $out
(
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
python tools/window_timing.py cfg5 1000 8
python tools/window_timing.py cfg3 60000 8
for b in 4096 65536; do BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b; done
BENCH_Q=8 python tools/bench_cfg.py cfg2 100000 4096
) > $out/log.txt 2>&1
grep -v amdgpu.ids $out/log.txt | tail -40

#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/window_test.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
for k in 1 2; do for w in 12 0; do
echo -n "KGPU_WINDOW=$w " | tee -a $OUT; KGPU_WINDOW=$w timeout 300 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT
echo -n "KGPU_WINDOW=$w " | tee -a $OUT; KGPU_WINDOW=$w timeout 300 python tools/bench_cfg.py cfg5 5000 1000 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT
done; done
bash tools/ab.sh 2 libkanpyo_gpu.so | tee -a $OUT

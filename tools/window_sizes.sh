#!/bin/bash
# windowed kernel: LDS per workgroup and positions per window on cfg 5 -> gpurun_out/window_sizes.txt
mkdir -p gpurun_out; OUT=gpurun_out/window_sizes.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8 KGPU_WINDOW_TRACE=1
L=$PWD/kanpyo_amd
run() { echo -n "$1 KGPU_WINDOW=$2: " | tee -a $OUT; KGPU_LIB=$L/$1 KGPU_WINDOW=$2 timeout 300 python tools/bench_cfg.py cfg5 5000 1000 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT; }
for kib in 11 12 13 14; do run libkanpyo_gpu.so $kib; done
for kib in 14 16 20; do run libkanpyo_gpu_win48.so $kib; done
for kib in 16 20 24; do run libkanpyo_gpu_win64.so $kib; done
timeout 700 python tools/fuzz_parity.py 500 424242 > gpurun_out/fuzz_r03_c.log 2>&1; tail -1 gpurun_out/fuzz_r03_c.log | tee -a $OUT

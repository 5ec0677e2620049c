#!/bin/bash
# cfg 3 with the windowed kernel taking shorter sentences too (build parameter KGPU_WINDOW_MIN_BYTES) -> gpurun_out/window_min.txt
mkdir -p gpurun_out; OUT=gpurun_out/window_min.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
L=$PWD/kanpyo_amd
run() { echo -n "$1 KGPU_WINDOW=$2: " | tee -a $OUT; KGPU_LIB=$L/$1 KGPU_WINDOW=$2 timeout 300 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT; }
run libkanpyo_gpu.so 12
run libkanpyo_gpu_wmin600.so 12
run libkanpyo_gpu_wmin600.so 16
run libkanpyo_gpu_wmin1000.so 12
run libkanpyo_gpu_wmin1000.so 16
run libkanpyo_gpu.so 12

#!/bin/bash
# windowed kernel, sustained rate (8 launches in flight): positions per window x LDS per workgroup on cfg 5 -> gpurun_out/window_sizes2.txt
mkdir -p gpurun_out; OUT=gpurun_out/window_sizes2.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8
L=$PWD/kanpyo_amd
run() { echo -n "$1 KGPU_WINDOW=$2: " | tee -a $OUT; KGPU_LIB=$L/$1 KGPU_WINDOW=$2 timeout 300 python tools/window_timing.py cfg5 1000 8 2>&1 | tail -3 | head -1 | cut -c1-200 | tee -a $OUT; }
run libkanpyo_gpu.so 12
run libkanpyo_gpu_win20.so 10
run libkanpyo_gpu_win24.so 10
run libkanpyo_gpu_win24.so 11
run libkanpyo_gpu_win28.so 11
run libkanpyo_gpu_win28.so 12
run libkanpyo_gpu.so 12

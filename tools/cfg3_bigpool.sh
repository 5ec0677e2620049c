#!/bin/bash
# cfg 3 (8-512 chars): one big shared pool per CU instead of 40 KB pools + the long-sentence kernel -> gpurun_out/cfg3_bigpool.txt
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
mkdir -p gpurun_out; OUT=gpurun_out/cfg3_bigpool.txt; : > $OUT
for pool in "40:4:48" "80:8:64" "160:16:64" "160:12:64" "160:8:64" "80:8:32,160:8:64"; do
  echo -n "KGPU_POOL=$pool : " | tee -a $OUT; KGPU_POOL=$pool timeout 200 python tools/bench_cfg.py cfg3 200000 2>&1 | tail -1 | tee -a $OUT
done

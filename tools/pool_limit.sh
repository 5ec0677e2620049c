#!/bin/bash
# cfg 3: pages of a pool (of 64) beyond which a sentence takes the long-sentence kernels -> gpurun_out/pool_limit.txt
mkdir -p gpurun_out; OUT=gpurun_out/pool_limit.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
for mp in 48 16 24 32 40 48; do
echo -n "KGPU_POOL=40:4:$mp " | tee -a $OUT; KGPU_POOL=40:4:$mp timeout 300 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT
done

#!/bin/bash
# cfg 3 / cfg 5: is it the launch's tail (one sentence per wavefront slot, the launch lasts as long as its longest sentence)?  Larger batches.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out; OUT=gpurun_out/cfg3_batch.txt; : > $OUT
for q in 8 16; do for b in 4096 16384 65536; do for pool in "40:4:48" "160:16:64" "80:8:64"; do
  echo -n "Q=$q batch=$b KGPU_POOL=$pool : " | tee -a $OUT; BENCH_Q=$q KGPU_POOL=$pool timeout 200 python tools/bench_cfg.py cfg3 400000 $b 2>&1 | tail -1 | tee -a $OUT
done; done; done
for b in 250 1000 4000; do
  echo -n "Q=8 batch=$b cfg5 : " | tee -a $OUT; BENCH_Q=8 timeout 200 python tools/bench_cfg.py cfg5 8000 $b 2>&1 | tail -1 | tee -a $OUT
done

#!/bin/bash
# sweep of batch size x batches in flight x pool shape (tools/resident_probe.py) -> gpurun_out/resident_probe.txt
mkdir -p gpurun_out
OUT=gpurun_out/resident_probe.txt; : > $OUT
run() { timeout 300 python tools/resident_probe.py "$@" 2>/dev/null | tail -1 | tee -a $OUT; }
for pool in 40:4:48 80:8:48 160:16:48; do
  export KGPU_POOL=$pool
  run 4096 8
  run 16384 4
  run 102400 2
  run 102400 1
done
export KGPU_POOL=40:4:64; run 102400 2
export KGPU_POOL=160:16:64; run 102400 2

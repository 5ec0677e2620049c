#!/bin/bash
export KGPU_TIERS=0
for cfg in "8 5" "8 6" "8 7" "16 8" "16 12" "2 1" "2 2" "3 2" "3 3" "6 5"; do
  set -- $cfg
  echo "== hwq $1 Q $2"
  GPU_MAX_HW_QUEUES=$1 KGPU_POOL="80:8" BENCH_Q=$2 timeout 120 python tools/bench_cfg.py cfg2 98304 4096 2>&1 | tail -1
done

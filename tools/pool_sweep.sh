#!/bin/bash
# Throughput of cfg2 under different LDS pool shapes ("KiB:wavefronts[,KiB:wavefronts]") and numbers of
# batches in flight.  usage (GPU box): bash tools/pool_sweep.sh
for p in "80:8" "160:16" "40:4" "80:8,160:4"; do
  for q in 2 3 4; do
    echo "== pool $p, $q batches in flight"
    KGPU_POOL="$p" BENCH_Q=$q timeout 120 python tools/bench_cfg.py cfg2 98304 4096 2>&1 | tail -1
  done
  KGPU_POOL="$p" BENCH_Q=2 timeout 120 python tools/bench_cfg.py cfg2 131072 65536 2>&1 | tail -1
done

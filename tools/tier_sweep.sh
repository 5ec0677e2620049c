#!/bin/bash
# throughput of cfg2 under different LDS tier chains ("KiB[:workgroups per CU]"), one big batch and 4096 x Q in flight
for t in "14:8,160" "14:9,160" "14:10,160" "14:11,160" "12:12,20,160" "12:13,20,160" "12:10,20,160" "12:11,20,160" "10:16,20,160" "10:12,20,160" "10:14,20,160"; do
  echo "== tiers $t"
  KGPU_TIERS="$t" BENCH_Q=2 timeout 120 python tools/bench_cfg.py cfg2 131072 65536 2>&1 | tail -1
  KGPU_TIERS="$t" BENCH_Q=8 timeout 120 python tools/bench_cfg.py cfg2 98304 4096 2>&1 | tail -1
done

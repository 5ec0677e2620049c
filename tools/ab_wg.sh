#!/bin/bash
# A/B of KGPU_POOL_WG (workgroups per pool launch) with the current library: bash tools/ab_wg.sh <reps> <n>...
REPS=$1; shift
for r in $(seq $REPS); do for wg in "$@"; do
  v=$(KGPU_POOL_WG=$wg timeout 200 python bench.py --no-cpu --no-extras ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2))")
  echo "$wg $v"
done; done | sort -n

#!/bin/bash
# A/B of KGPU_POOL plans with the current library: bash tools/ab_pool.sh <reps> <plan>...
REPS=$1; shift
for r in $(seq $REPS); do for pool in "$@"; do
  v=$(KGPU_POOL=$pool timeout 200 python bench.py --no-cpu --no-extras ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2), d['routing']['redone'][0], d['routing']['deferred'][0])")
  echo "$pool $v"
done; done | sort

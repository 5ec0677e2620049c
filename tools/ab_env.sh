#!/bin/bash
# A/B of (library, KGPU_POOL) pairs on the GPU box: bash tools/ab_env.sh <reps> <lib>@<pool> ...   (bench.py headline per run, interleaved)
REPS=$1; shift
for r in $(seq $REPS); do for cfg in "$@"; do
  lib=${cfg%@*}; pool=${cfg#*@}
  v=$(KGPU_LIB=$PWD/kanpyo_amd/$lib KGPU_POOL=$pool timeout 200 python bench.py --no-cpu --no-extras ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['value']/1e6,2))")
  echo "$cfg $v"
done; done | sort | awk '{a[$1]=a[$1]" "$2} END {for (k in a) print k":"a[k]}'

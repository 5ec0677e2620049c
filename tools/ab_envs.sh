#!/bin/bash
# A/B of environment settings with the current library: bash tools/ab_envs.sh <reps> "VAR=val VAR2=val" ...   ("-" = none)
REPS=$1; shift
for r in $(seq $REPS); do for e in "$@"; do
  v=$(env $( [ "$e" = "-" ] || echo $e ) timeout 200 python bench.py --no-cpu --no-extras ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2), d['routing']['redone'][0])")
  echo "$e: $v"
done; done | sort

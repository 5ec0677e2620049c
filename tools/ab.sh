#!/bin/bash
# A/B of library builds on the GPU box: bash tools/ab.sh <reps> <lib>...   (bench.py headline value per run, same box, interleaved)
REPS=$1; shift
for r in $(seq $REPS); do for lib in "$@"; do
  v=$(KGPU_LIB=$PWD/kanpyo_amd/$lib timeout 200 python bench.py --no-cpu --no-extras ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['value']/1e6,2))")
  echo "$lib $v"
done; done | sort | awk '{a[$1]=a[$1]" "$2} END {for (k in a) print k":"a[k]}'

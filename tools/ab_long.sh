#!/bin/bash
# A/B of library builds on the long-sentence configs: bash tools/ab_long.sh <reps> <lib>...
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
REPS=$1; shift
for r in $(seq $REPS); do for lib in "$@"; do
  a=$(KGPU_LIB=$PWD/kanpyo_amd/$lib timeout 200 python tools/bench_cfg.py cfg3 400000 16384 2>/dev/null | tail -1 | sed 's/.*avg: \([0-9,]*\) sentences.*/\1/')
  b=$(KGPU_LIB=$PWD/kanpyo_amd/$lib timeout 200 python tools/bench_cfg.py cfg5 5000 1000 2>/dev/null | tail -1 | sed 's/.*avg: \([0-9,]*\) sentences.*/\1/')
  echo "$lib cfg3 $a cfg5 $b"
done; done

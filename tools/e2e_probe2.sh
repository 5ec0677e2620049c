#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/e2e_probe2.txt; : > $OUT
export KGPU_TEST_HOOKS_REREAD=1
run() { timeout 300 python tools/e2e_probe.py "$@" 2>&1 | grep -E "^e2e|kgpu_tokenize_batch:" | tail -2 | tee -a $OUT; }
KGPU_HOST_TRACE=1 run 4 pinned
KGPU_HOST_TRACE=1 run 4
run 4 pinned
run 4
run 1 pinned
run 1
KGPU_HOST_DEPTH=16 run 4 pinned
KGPU_HOST_DEPTH=8 run 4 pinned

#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/e2e_eng.txt; : > $OUT
run() { timeout 300 python tools/e2e_probe.py "$@" 2>&1 | grep -E "^e2e" | tail -1 | tee -a $OUT; }
echo plain | tee -a $OUT; run 4 pinned; run 4
echo "with 8 contexts on 4 torch streams" | tee -a $OUT; PROBE_ENG=1 run 4 pinned; PROBE_ENG=1 run 4
echo "torch streams only" | tee -a $OUT; PROBE_ENG=2 run 4 pinned

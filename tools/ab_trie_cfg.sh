#!/bin/bash
# character-level trie against the byte-level walk on cfg 3 / cfg 5 (same library, KGPU_BYTE_TRIE=1: no char-level copy) -> gpurun_out/ab_trie_cfg.txt
mkdir -p gpurun_out; OUT=gpurun_out/ab_trie_cfg.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
for r in 1 2; do for bt in 0 1; do
echo -n "KGPU_BYTE_TRIE=$bt " | tee -a $OUT; KGPU_BYTE_TRIE=$bt timeout 300 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT
echo -n "KGPU_BYTE_TRIE=$bt " | tee -a $OUT; KGPU_BYTE_TRIE=$bt KGPU_WINDOW=12 timeout 300 python tools/window_timing.py cfg5 1000 8 2>&1 | tail -3 | head -2 | cut -c1-300 | tee -a $OUT
done; done

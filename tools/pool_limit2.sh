#!/bin/bash
# pages of a pool (of 64) beyond which a sentence takes the long-sentence kernels, with the character-level trie: cfg 2 and cfg 3 -> gpurun_out/pool_limit2.txt
mkdir -p gpurun_out; OUT=gpurun_out/pool_limit2.txt; : > $OUT
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
for r in 1 2; do for mp in 48 40 32; do
v=$(KGPU_POOL=40:4:$mp timeout 200 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['value']/1e6,2))")
echo -n "KGPU_POOL=40:4:$mp cfg2 $v  " | tee -a $OUT; KGPU_POOL=40:4:$mp timeout 300 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | cut -c1-120 | tee -a $OUT
done; done

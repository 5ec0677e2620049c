#!/bin/bash
# long-sentence kernel: LDS per workgroup (= resident workgroups per CU) in the saturated regime -> gpurun_out/long_lds.txt
export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
mkdir -p gpurun_out; OUT=gpurun_out/long_lds.txt; : > $OUT
for kib in 12 8 10 16 24; do
  echo -n "KGPU_LONG=$kib cfg3 batch 16384: " | tee -a $OUT; KGPU_LONG=$kib timeout 200 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | tee -a $OUT
  echo -n "KGPU_LONG=$kib cfg5 batch 1000: " | tee -a $OUT; KGPU_LONG=$kib timeout 200 python tools/bench_cfg.py cfg5 5000 1000 2>&1 | tail -1 | tee -a $OUT
done
export KGPU_TEST_HOOKS_REREAD=1
for k in 1 2; do timeout 300 python tools/e2e_probe.py 4 pinned 2>&1 | grep "^e2e" | tee -a $OUT; timeout 300 python tools/e2e_probe.py 4 2>&1 | grep "^e2e" | tee -a $OUT; done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT

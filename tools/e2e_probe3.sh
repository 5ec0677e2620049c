#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/e2e_probe3.txt; : > $OUT
export KGPU_TEST_HOOKS_REREAD=1
run() { timeout 300 python tools/e2e_probe.py "$@" 2>&1 | grep -E "^e2e|kgpu_tokenize_batch:" | tail -2 | tee -a $OUT; }
for k in 1 2; do
echo "xfer stream:" | tee -a $OUT
KGPU_HOST_TRACE=1 run 4 pinned
run 4
run 1 pinned
echo "no xfer stream:" | tee -a $OUT
KGPU_NO_XFER_STREAM=1 KGPU_HOST_TRACE=1 run 4 pinned
KGPU_NO_XFER_STREAM=1 run 4
KGPU_NO_XFER_STREAM=1 run 1 pinned
done
echo "without GPU_MAX_HW_QUEUES:" | tee -a $OUT
GPU_MAX_HW_QUEUES=4 run 4 pinned
GPU_MAX_HW_QUEUES=4 KGPU_NO_XFER_STREAM=1 run 4 pinned

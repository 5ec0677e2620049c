#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/repro.txt; : > $OUT
export KGPU_TEST_HOOKS_REREAD=1
TAG=default timeout 900 python tools/status_repro.py 1000000 2>&1 | tail -5 | tee -a $OUT
TAG=legacy KGPU_HOST_LEGACY=1 timeout 900 python tools/status_repro.py 1000000 2>&1 | tail -5 | tee -a $OUT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 | tee -a $OUT

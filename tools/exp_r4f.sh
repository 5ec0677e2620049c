#!/bin/bash
cd /root/repo
out=gpurun_out/r4f; mkdir -p $out
(
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
echo "== cfg5 / cfg3-long phases (main lib)"
python tools/window_timing.py cfg5 1000 8
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so python tools/window_timing.py cfg3 60000 8
echo "== cfg3 full mix, 400k sentences"
for b in 4096 65536; do
BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so KGPU_LONG=0 BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1w5.so KGPU_WINDOW=10 BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
done
) > $out/log.txt 2>&1
grep -v amdgpu.ids $out/log.txt | tail -60

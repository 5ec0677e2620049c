#!/bin/bash
cd /root/repo
out=gpurun_out/r4g; mkdir -p $out
(
for b in 4096 65536; do
BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so KGPU_LONG=0 BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so KGPU_POOL=40:4:48 BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m1.so KGPU_POOL=40:4:32 BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
done
) > $out/log.txt 2>&1
grep -v amdgpu.ids $out/log.txt | tail -60

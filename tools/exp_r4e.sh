#!/bin/bash
# round-4 experiment: windowed-kernel variants (occupancy vs window size), concurrent small calls traced
cd /root/repo
out=gpurun_out/r4e; mkdir -p $out
(
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "window or cfg5 or long" 2>&1 | tail -3
echo "== cfg5"
KGPU_WINDOW=12 python tools/window_timing.py cfg5 1000 8
for kb in 8 9 10 12; do KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w5.so KGPU_WINDOW=$kb python tools/window_timing.py cfg5 1000 8; done
for kb in 7 8 10; do KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w5n24.so KGPU_WINDOW=$kb python tools/window_timing.py cfg5 1000 8; done
for kb in 6 7 8; do KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w6n16.so KGPU_WINDOW=$kb python tools/window_timing.py cfg5 1000 8; done
echo "== cfg3 long sentences"
KGPU_WINDOW=12 python tools/window_timing.py cfg3 60000 8
for kb in 12 16; do KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m600.so KGPU_WINDOW=$kb python tools/window_timing.py cfg3 60000 8; done
for kb in 8 10 12; do KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_m600w5.so KGPU_WINDOW=$kb python tools/window_timing.py cfg3 60000 8; done
echo "== concurrent"
KGPU_SMALL_TRACE=1 python tools/concurrent_probe.py 1,4,16,64
) > $out/log.txt 2>&1
grep -v amdgpu.ids $out/log.txt | tail -80

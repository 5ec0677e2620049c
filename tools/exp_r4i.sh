#!/bin/bash
cd /root/repo
out=gpurun_out/r4i; mkdir -p $out
(
python tools/clock_probe.py
python tools/concurrent_probe.py 16,64,256
for b in 65536 4096; do
for pool in 40:4:40 40:4:48 40:4:64; do for w in 12 16; do
echo "pool $pool window $w batch $b"; KGPU_POOL=$pool KGPU_WINDOW=$w BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 $b
done; done; done
) > $out/log.txt 2>&1
grep -v amdgpu.ids $out/log.txt | tail -60

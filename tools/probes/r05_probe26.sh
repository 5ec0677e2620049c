#!/bin/bash
# The combiner's lock as a spinlock: 128 / 64 / 256 / 16 / 1 callers, CPU per call and the cgroup's throttling (before: tools/probes/_ab/libkanpyo_gpu_before.so, pthread mutex)
REPO=$(cd "$(dirname "$0")/../.." && pwd); cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth.pkl
OUT=gpurun_out/p26; mkdir -p $OUT
{
echo "== 128 threads, traced"; KGPU_SMALL_TRACE=1 timeout 300 python tools/callers_probe.py 128 200 4
echo "== 128 threads"; timeout 300 python tools/callers_probe.py 128 200 10
echo "== 64 threads";  timeout 300 python tools/callers_probe.py 64 300 6
echo "== 256 threads"; timeout 300 python tools/callers_probe.py 256 100 4
echo "== 16 threads"; timeout 300 python tools/callers_probe.py 16 300 4
echo "== 1 thread"; timeout 300 python tools/callers_probe.py 1 400 3
echo "== 16 threads, before"; KGPU_LIB=$REPO/tools/probes/_ab/libkanpyo_gpu_before.so timeout 300 python tools/callers_probe.py 16 300 4
echo "== 64 threads, before"; KGPU_LIB=$REPO/tools/probes/_ab/libkanpyo_gpu_before.so timeout 300 python tools/callers_probe.py 64 300 4
echo "== 128 threads, pinned to 16 CPUs (affinity instead of quota: a preempted lock holder)"; timeout 300 taskset -c 0-15 python tools/callers_probe.py 128 200 4
echo "== 128 threads, pinned to 16 CPUs, before"; KGPU_LIB=$REPO/tools/probes/_ab/libkanpyo_gpu_before.so timeout 300 taskset -c 0-15 python tools/callers_probe.py 128 200 4
} > $OUT/callers_spin.txt 2>&1
grep -v amdgpu.ids $OUT/callers_spin.txt | tail -90

#!/bin/bash
# The 128-thread callers' slow mode (p99 58-69 ms in two of four collections): is it the cgroup quota, and is it the measuring harness's start gate?
# before = the library of commit f0a3f43 (threads spin on yield until all are started), after = the futex gate.
REPO=$(cd "$(dirname "$0")/../.." && pwd); cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth.pkl
OUT=gpurun_out/p24; mkdir -p $OUT
{
for arm in before after before after; do
  if [ $arm = before ]; then export KGPU_LIB=$REPO/tools/probes/_ab/libkanpyo_gpu_before.so; else unset KGPU_LIB; fi
  echo "== $arm"
  timeout 300 python tools/callers_probe.py 128 200 10
done
unset KGPU_LIB
echo "== after, 64 threads"
timeout 300 python tools/callers_probe.py 64 300 8
} > $OUT/callers.txt 2>&1
tail -70 $OUT/callers.txt

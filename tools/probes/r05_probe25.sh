#!/bin/bash
# Where the CPU time of 128 / 64 one-sentence callers goes (KGPU_SMALL_TRACE: thread CPU clocks at the combiner's phase boundaries), and the cgroup's user / system split.
REPO=$(cd "$(dirname "$0")/../.." && pwd); cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth.pkl
OUT=gpurun_out/p25; mkdir -p $OUT
{
echo "== 128 threads, traced"; KGPU_SMALL_TRACE=1 timeout 300 python tools/callers_probe.py 128 200 6
echo "== 64 threads, traced";  KGPU_SMALL_TRACE=1 timeout 300 python tools/callers_probe.py 64 300 4
echo "== 256 threads"; timeout 300 python tools/callers_probe.py 256 100 4
echo "== 128 threads, KGPU_CPU_BUDGET=1000 (never the crowded mode)"; KGPU_CPU_BUDGET=1000 timeout 300 python tools/callers_probe.py 128 200 6
} > $OUT/callers_cpu.txt 2>&1
tail -80 $OUT/callers_cpu.txt

#!/bin/bash
# The combiner's lock with backoff, its hot words on cache lines of their own; short runs as in bench.py and SUSTAINED ones (several 100 ms quota periods)
REPO=$(cd "$(dirname "$0")/../.." && pwd); cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth.pkl
OUT=gpurun_out/p27; mkdir -p $OUT
{
echo "== 128 threads, traced"; KGPU_SMALL_TRACE=1 timeout 300 python tools/callers_probe.py 128 200 4
echo "== 128 threads"; timeout 300 python tools/callers_probe.py 128 200 8
echo "== 128 threads, sustained (3000 calls each)"; timeout 300 python tools/callers_probe.py 128 3000 3
echo "== 64 threads";  timeout 300 python tools/callers_probe.py 64 300 4
echo "== 64 threads, sustained (6000 calls each)";  timeout 300 python tools/callers_probe.py 64 6000 3
echo "== 256 threads"; timeout 300 python tools/callers_probe.py 256 100 4
echo "== 16 threads"; timeout 300 python tools/callers_probe.py 16 300 4
echo "== 128 threads sustained, before (pthread mutex)"; KGPU_LIB=$REPO/tools/probes/_ab/libkanpyo_gpu_before.so timeout 300 python tools/callers_probe.py 128 3000 2
echo "== 64 threads sustained, before (pthread mutex)"; KGPU_LIB=$REPO/tools/probes/_ab/libkanpyo_gpu_before.so timeout 300 python tools/callers_probe.py 64 6000 2
} > $OUT/callers_spin2.txt 2>&1
grep -v amdgpu.ids $OUT/callers_spin2.txt | tail -90

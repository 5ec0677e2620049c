#!/bin/bash
# round 5: host-side knobs once more on 16 hardware queues -- worker threads of the large host call, the combiner's launches in flight / streams
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p22; mkdir -p "$O"
{
for t in 8 12 16; do echo "== KGPU_HOST_THREADS=$t"; KGPU_HOST_THREADS=$t timeout 300 python tools/e2e_quick.py 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== callers: default"; timeout 300 python tools/concurrent_probe.py 16,64,128 2>&1 | grep -v amdgpu.ids
echo "== callers: KGPU_COMBINE_LAUNCHES=8"; KGPU_COMBINE_LAUNCHES=8 timeout 300 python tools/concurrent_probe.py 16,64,128 2>&1 | grep -v amdgpu.ids
echo "== callers: KGPU_COMBINE_LAUNCHES=8 KGPU_STREAMS=8"; KGPU_COMBINE_LAUNCHES=8 KGPU_STREAMS=8 timeout 300 python tools/concurrent_probe.py 16,64,128 2>&1 | grep -v amdgpu.ids
echo "== callers: KGPU_COMBINE_LAUNCHES=2"; KGPU_COMBINE_LAUNCHES=2 timeout 300 python tools/concurrent_probe.py 64,128 2>&1 | grep -v amdgpu.ids
echo "== callers: KGPU_COMBINE_US=24"; KGPU_COMBINE_US=24 timeout 300 python tools/concurrent_probe.py 64,128 2>&1 | grep -v amdgpu.ids
} > "$O/host_knobs.txt" 2>&1
cat "$O/host_knobs.txt"

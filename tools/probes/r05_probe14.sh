#!/bin/bash
# round 5: the pool kernel's workgroup shape behind... in front of a windowed launch (cfg 3): smaller workgroups find a place sooner on a chip full of 10 KB single-wavefront workgroups
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p14; mkdir -p "$O"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for pool in 40:4:24 40:4:32 20:2:48 20:2:64 30:3:40 10:1:64 80:8:12; do
  c BENCH_Q=8 KGPU_POOL=$pool python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_POOL=$pool python tools/bench_cfg.py cfg3 400000 65536
done
} > "$O/pool_shapes.txt" 2>&1
cat "$O/pool_shapes.txt"

#!/bin/bash
# round 5: two-wavefront pools of 20 KB for every workload?  (cfg 3 gained 16 % / 5 % with them in probe 14)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p17; mkdir -p "$O"
bash tools/ab.sh -r 2 -c bench p40:KGPU_POOL=40:4:32 p20_64:KGPU_POOL=20:2:64 p20_48:KGPU_POOL=20:2:48 p10:KGPU_POOL=10:1:64 > "$O/ab_cfg2.txt" 2>&1
cat "$O/ab_cfg2.txt"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for pool in 40:4:32 20:2:64 20:2:48 10:1:64; do
  c BENCH_Q=8 KGPU_POOL=$pool python tools/bench_cfg.py dense 200000 4096
  c BENCH_Q=1 KGPU_POOL=$pool python tools/bench_cfg.py cfg2 400000 4096
done
for pool in 20:2:40 20:2:56 16:2:64 24:2:64; do
  c BENCH_Q=8 KGPU_POOL=$pool python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_POOL=$pool python tools/bench_cfg.py cfg3 400000 65536
done
} > "$O/others.txt" 2>&1
cat "$O/others.txt"

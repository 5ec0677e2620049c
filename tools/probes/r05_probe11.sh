#!/bin/bash
# round 5: the pool's routing limit once more (lower), the dense leg under it, the split chain
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p11; mkdir -p "$O"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for mp in 16 20 24 28; do
  c BENCH_Q=8 KGPU_POOL=40:4:$mp python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_POOL=40:4:$mp python tools/bench_cfg.py cfg3 400000 65536
done
for mp in 24 32 40; do
  c BENCH_Q=8 KGPU_POOL=40:4:$mp python tools/bench_cfg.py dense 200000 4096
  c BENCH_Q=8 KGPU_POOL=40:4:$mp python tools/bench_cfg.py cfg2 400000 4096
done
c BENCH_Q=8 KGPU_SPLIT_CHAIN=1 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 KGPU_SPLIT_CHAIN=1 python tools/bench_cfg.py cfg3 400000 65536
c BENCH_Q=8 KGPU_SPLIT_CHAIN=1 KGPU_POOL=40:4:24 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 KGPU_SPLIT_CHAIN=1 KGPU_POOL=40:4:24 python tools/bench_cfg.py cfg3 400000 65536
} > "$O/sweep2.txt" 2>&1
cat "$O/sweep2.txt"

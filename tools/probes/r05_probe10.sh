#!/bin/bash
# round 5: cfg 3 on the new stream arrangement -- the pool's routing limit, the number of long streams, contexts
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p10; mkdir -p "$O"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for mp in 24 32 40 48; do
  c BENCH_Q=8 KGPU_POOL=40:4:$mp python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_POOL=40:4:$mp python tools/bench_cfg.py cfg3 400000 65536
done
c BENCH_Q=8 KGPU_LONG_STREAMS=4 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=12 KGPU_LONG_STREAMS=12 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=16 KGPU_LONG_STREAMS=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 KGPU_WINDOW=12 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 KGPU_WINDOW=12 python tools/bench_cfg.py cfg3 400000 65536
} > "$O/cfg3_sweep.txt" 2>&1
cat "$O/cfg3_sweep.txt"

#!/bin/bash
# round 5: the windowed launch's grid behind the pools by the host's estimate (KGPU_WINDOW_GRID=0: the full 4096 as before), then the pool-shape sweep of probe 14
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p15; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for r in 1 2; do for g in 0 -1; do
  c BENCH_Q=8 KGPU_WINDOW_GRID=$g python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_WINDOW_GRID=$g python tools/bench_cfg.py cfg3 400000 16384
done; done
c BENCH_Q=8 KGPU_WINDOW_GRID=0 python tools/bench_cfg.py cfg2 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py cfg2 400000 4096
} > "$O/window_grid.txt" 2>&1
cat "$O/window_grid.txt"
bash tools/probes/r05_probe14.sh 2>&1 | tail -30

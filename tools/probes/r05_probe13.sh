#!/bin/bash
# round 5: scan / compaction in small workgroups behind chains with a windowed launch (KGPU_SCAN_SMALL: 0 never, 1 always, unset = by the chain)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p13; mkdir -p "$O"
KGPU_SCAN_SMALL=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_team.py -m gpu -x -q 2>&1 | tail -2
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for r in 1 2; do for ss in 0 1; do
  c KGPU_SCAN_SMALL=$ss python tools/team_time.py
  c BENCH_Q=8 KGPU_SCAN_SMALL=$ss python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_SCAN_SMALL=$ss python tools/bench_cfg.py cfg3 400000 65536
done; done
c BENCH_Q=8 KGPU_SCAN_SMALL=1 python tools/bench_cfg.py cfg2 400000 4096
c BENCH_Q=8 KGPU_SCAN_SMALL=0 python tools/bench_cfg.py cfg2 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 4096
c python tools/team_time.py
} > "$O/scan_small.txt" 2>&1
cat "$O/scan_small.txt"

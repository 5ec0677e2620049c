#!/bin/bash
# round 5: scan + compaction in one launch (KGPU_SCAN_COMPACT=1) behind chains with a windowed launch -- the two small kernels wait for a CU behind long-running workgroups
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p12; mkdir -p "$O"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for sc in 0 1; do
  c BENCH_Q=8 KGPU_SCAN_COMPACT=$sc python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 KGPU_SCAN_COMPACT=$sc python tools/bench_cfg.py cfg3 400000 65536
  c KGPU_SCAN_COMPACT=$sc python tools/team_time.py
done
c BENCH_Q=8 KGPU_SCAN_COMPACT=1 KGPU_SCAN_WG=256 python tools/bench_cfg.py cfg3 400000 4096
} > "$O/scan_compact.txt" 2>&1
cat "$O/scan_compact.txt"

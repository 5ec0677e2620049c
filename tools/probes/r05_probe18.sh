#!/bin/bash
# round 5: the shipped defaults after the per-chain pool shape, the windowed launch's estimated grid and the small scan / compaction workgroups
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p18; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; tail -3 "$O/pytest_gpu.txt"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 16384
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 65536
c BENCH_Q=8 python tools/bench_cfg.py cfg2 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py dense 200000 4096
c python tools/team_time.py
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 65536
} > "$O/defaults.txt" 2>&1
cat "$O/defaults.txt"
timeout 700 python tools/fuzz_parity.py 600 20261104 > "$O/fuzz.txt" 2>&1; tail -1 "$O/fuzz.txt"

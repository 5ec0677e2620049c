#!/bin/bash
# round 5: the gpu tests on the new runtime (window-first chains, long streams, 16 hardware queues), then the three workloads with the defaults
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p6; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.txt" 2>&1
tail -5 "$O/pytest_gpu.txt"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
c python tools/window_timing.py cfg5 1000 8
c python tools/window_timing.py cfg5 1000 1
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 65536
c BENCH_Q=8 python tools/bench_cfg.py cfg2 400000 4096
} > "$O/defaults.txt" 2>&1
python tools/window_timing.py cfg5 1000 8 2>&1 | grep -v amdgpu.ids | head -3 >> "$O/defaults.txt"
cat "$O/defaults.txt"
echo done

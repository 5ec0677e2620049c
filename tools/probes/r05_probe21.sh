#!/bin/bash
# round 5: the windowed kernel's ordinary form claiming its sentences one by one (shipped build) against the static every-G-th form (-DKGPU_WIN_STATIC)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p21; mkdir -p "$O"
ST=$REPO/kanpyo_amd/libkanpyo_gpu_static.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_team.py -m gpu -x -q 2>&1 | tail -2
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for r in 1 2; do for lib in "A=1" "KGPU_LIB=$ST"; do
  c BENCH_Q=8 $lib python tools/bench_cfg.py cfg3 400000 65536
  c BENCH_Q=8 $lib python tools/bench_cfg.py cfg3 400000 16384
  c BENCH_Q=8 $lib python tools/bench_cfg.py cfg3 400000 4096
  c $lib python tools/team_time.py
done; done
} > "$O/win_ticket.txt" 2>&1
cat "$O/win_ticket.txt"

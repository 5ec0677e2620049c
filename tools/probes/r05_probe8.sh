#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p8; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1
tail -15 "$O/pytest_gpu.txt"
timeout 400 python tools/fuzz_parity.py 300 20261101 > "$O/fuzz.txt" 2>&1
tail -3 "$O/fuzz.txt"
timeout 300 python tools/team_time.py 2>&1 | grep -v amdgpu.ids > "$O/team_time.txt"
cat "$O/team_time.txt"

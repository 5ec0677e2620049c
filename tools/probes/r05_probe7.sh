#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p7; mkdir -p "$O"
timeout 900 python tools/team_check.py quick > "$O/team_check.txt" 2>&1
tail -30 "$O/team_check.txt"
{ KGPU_WINDOW_TEAM=0 timeout 300 python tools/team_time.py; KGPU_WINDOW_TEAM=2 timeout 300 python tools/team_time.py; timeout 300 python tools/team_time.py; } 2>&1 | grep -v amdgpu.ids > "$O/team_time.txt"
cat "$O/team_time.txt"

#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p19; mkdir -p "$O"
bash tools/ab.sh -r 3 -c bench plain scan256:KGPU_SCAN_SMALL=2 > "$O/ab_cfg2.txt" 2>&1; cat "$O/ab_cfg2.txt"
timeout 200 python tools/team_time.py 2>&1 | grep -v amdgpu.ids | head -2

#!/bin/bash
# round 5, the final tree: the gpu tests, a long fuzz run, then everything profiles/ holds for the round (tools/collect_profiles.sh)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/final; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; tail -4 "$O/pytest_gpu.txt"
timeout 1000 python tools/fuzz_parity.py ${FUZZ_SECONDS:-900} 20261103 > "$O/fuzz.txt" 2>&1; tail -2 "$O/fuzz.txt"
bash tools/collect_profiles.sh gpurun_out/r05 r05 > "$O/collect.log" 2>&1; tail -2 "$O/collect.log"

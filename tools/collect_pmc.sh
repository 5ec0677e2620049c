#!/bin/bash
# GPU box: only the PMC passes of tools/collect_profiles.sh (cfg 2, pool kernel) -> <outdir>/<tag>_pmc_summary.{txt,json}, pmc_traffic.json, pmc_instructions.json
set -u
OUT=$(realpath -m "$1"); TAG=$2
REPO=$(cd "$(dirname "$0")/.." && pwd)
export GPU_MAX_HW_QUEUES=8
mkdir -p "$OUT"
bash "$REPO/tools/pmc_passes.sh" "$OUT/pmc" python "$REPO/bench.py" --steps 2 --warmup 1 --queue 1 --no-cpu --no-extras --no-stages > "$OUT/pmc.log" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT/pmc" --json "$OUT/${TAG}_pmc_summary.json" > "$OUT/${TAG}_pmc_summary.txt"
python "$REPO/tools/make_traffic_json.py" "$OUT/${TAG}_pmc_summary.json" "$OUT/pmc_traffic.json" "$TAG"
rm -rf "$OUT"/pmc/pass*/

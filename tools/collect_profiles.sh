#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + separate PMC passes of the bench command.
# usage: bash tools/collect_profiles.sh <outdir>
set -u
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
python "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$REPO/bench.py" --steps 96 --warmup 8 --no-cpu > "$OUT/trace.json" 2> "$OUT/trace.err"
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
bash "$REPO/tools/pmc_passes.sh" "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT/pmc" --json "$OUT/pmc_summary.json" > "$OUT/pmc_summary.txt"
python "$REPO/tools/make_traffic_json.py" "$OUT/pmc_summary.json" "$OUT/pmc_traffic.json" > /dev/null
rm -rf "$OUT/trace" "$OUT/pmc"/pass*/
echo done

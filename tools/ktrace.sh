#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of a short bench run (one context: kernels run alone) -> <outdir>/<tag>_kernel_stats.csv
# usage: bash tools/ktrace.sh <outdir> <tag> [bench args]
OUT=$(realpath -m "$1"); TAG=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$TAG" -- python "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu --no-extras "$@" > "$OUT/trace_$TAG.json" 2> "$OUT/trace_$TAG.err"
cp $(find "$OUT/trace_$TAG" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats.csv"
rm -rf "$OUT/trace_$TAG"
python - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}  {r['Percentage']}%")
PY

#!/bin/bash
# Per-phase counters of the pool kernel: early-stop ablation (kgpu_ctx_set_ablation via ABLATE_STOP=k) under one --pmc pass each.
# usage (GPU box): bash tools/pmc_phases.sh <outdir> [stops...]     KGPU_PMC_COUNTERS overrides the counter group
OUT=$(realpath -m "$1"); shift; REPO=$(cd "$(dirname "$0")/.." && pwd)
STOPS=${@:-1 2 3 4 5 6 7 0}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for k in $STOPS; do
  ABLATE_STOP=$k timeout 200 rocprofv3 --pmc ${KGPU_PMC_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU} \
     --output-format csv -d "$OUT/stop$k" -- python "$REPO/tools/ablate.py" child cfg2 4096 > "$OUT/stop$k.log" 2>&1
  echo "stop $k rc=$?"
done
python - "$OUT" $STOPS <<'PY'
import csv, glob, sys
from collections import defaultdict
names = {1: "load", 2: "decode", 3: "walk", 4: "scan", 5: "emit", 6: "gather", 7: "sweep", 0: "all"}  # (round 6: 5 = emit + tile list, 6 = the tile gathers alone, 7 = + the sweep)
prev = defaultdict(float)
stops = [int(x) for x in sys.argv[2:]]
first = True
for k in stops:
    tot = defaultdict(float); nd = defaultdict(set)
    for f in glob.glob(f"{sys.argv[1]}/stop{k}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_tokenize_pool" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]].add(r["Dispatch_Id"])
    per = {c: tot[c] / max(len(nd[c]), 1) / 4096 for c in tot}
    cols = tuple(sorted(per))
    if first:
        print("per sentence; each row = this stop level minus the previous one listed")
        print(f"{'phase':8s}" + "".join(f"{c[-16:]:>17s}" for c in cols)); first = False
    print(f"{names[k]:8s}" + "".join(f"{per.get(c, 0) - prev[c]:17.1f}" for c in cols))
    prev = defaultdict(float, per)
PY
rm -rf "$OUT"/stop*/

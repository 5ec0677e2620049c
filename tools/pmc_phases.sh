#!/bin/bash
# Per-phase instruction mix of the pool kernel: early-stop ablation (KGPU_DEBUG_STOP=k) under one --pmc pass each.
# usage (GPU box): bash tools/pmc_phases.sh <outdir>
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4 5 6 7 0; do
  KGPU_DEBUG_STOP=$k timeout 200 rocprofv3 --pmc ${KGPU_PMC_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU} \
     --output-format csv -d "$OUT/stop$k" -- python "$REPO/tools/ablate.py" child cfg2 4096 > "$OUT/stop$k.log" 2>&1
  echo "stop $k rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
names = {1: "load", 2: "decode", 3: "walk", 4: "scan", 5: "emit", 6: "gather", 7: "sweep", 0: "all"}
prev = defaultdict(float)
print(f"{'phase':8s}" + "".join(f"{c:>14s}" for c in ("VALU", "SALU", "LDS", "VMEM_RD", "ACTIVE_ANY", "ACTIVE_VALU", "WAIT_ANY", "WAVE_CYC")))
for k in (1, 2, 3, 4, 5, 6, 7, 0):
    tot = defaultdict(float); nd = defaultdict(set)
    for f in glob.glob(f"{sys.argv[1]}/stop{k}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_tokenize_pool" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 4096 * 64 // 8 * 8 // 8:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]].add(r["Dispatch_Id"])
    per = {c: tot[c] / max(len(nd[c]), 1) / 4096 for c in tot}
    cols = tuple(sorted(per)) if "SQ_INSTS_VALU" not in per else ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    if k == 1: print("columns:", cols)
    print(f"{names[k]:8s}" + "".join(f"{per.get(c, 0) - prev[c]:14.1f}" for c in cols))
    prev = defaultdict(float, per)
PY

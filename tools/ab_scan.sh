#!/bin/bash
# scan + compaction in one launch (default) against the separate scan kernel (KGPU_SCAN_COMPACT=2) -> gpurun_out/ab_scan.txt
mkdir -p gpurun_out; OUT=gpurun_out/ab_scan.txt; : > $OUT
for r in 1 2 3; do for m in 1 2; do
  v=$(KGPU_SCAN_COMPACT=$m timeout 200 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['value']/1e6,2))")
  echo "KGPU_SCAN_COMPACT=$m $v" | tee -a $OUT
done; done
for m in 1 2; do echo -n "KGPU_SCAN_COMPACT=$m " | tee -a $OUT; KGPU_SCAN_COMPACT=$m timeout 200 python tools/e2e_probe.py 4 2>&1 | grep -E "^e2e" | tail -1 | tee -a $OUT; done

#!/bin/bash
# device-resident path: separate scan + compaction kernels (default there) against the fused launch at 16 / 32 sentences per workgroup -> gpurun_out/ab_scan2.txt
mkdir -p gpurun_out; OUT=gpurun_out/ab_scan2.txt; : > $OUT
run() { v=$(env "$@" timeout 200 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['value']/1e6,2))"); echo "$* $v" | tee -a $OUT; }
for r in 1 2; do run KGPU_SCAN_COMPACT=0; run KGPU_SCAN_COMPACT=1 KGPU_SCAN_WG=16; run KGPU_SCAN_COMPACT=1 KGPU_SCAN_WG=32; done
for w in 64 32 128; do echo -n "host call, KGPU_SCAN_WG=$w " | tee -a $OUT; KGPU_SCAN_WG=$w timeout 200 python tools/e2e_probe.py 4 2>&1 | grep -E "^e2e" | tail -1 | tee -a $OUT; done

#!/bin/bash
# Extra PMC passes to find the saturated unit (GPU box). usage: bash tools/pmc_extra.sh <outdir> [bench args]
set -u
OUT=$(realpath -m "$1"); shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
  "SQ_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_LEVEL_WAVES" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python "$REPO/bench.py" --steps 24 --warmup 2 --no-cpu "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? : $grp"
done

#!/bin/bash
# Separate rocprofv3 --pmc passes (never combined with tracing; only the counter groups that have run clean on this pool --
# other groups have hung the profiler) over a short run of a command.
# usage (GPU box): bash tools/pmc_passes.sh <outdir> <command...>      e.g.  ... python bench.py --steps 2 --warmup 1 --queue 1 --no-cpu --no-extras
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? : $grp"
done

#!/bin/bash
# round 5, third GPU probe: VALU op classes, kernel concurrency, which unit of the CU is busy under the pool kernel (derived PMC metrics, one per pass)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p3; mkdir -p "$O"
./tools/ubench/valu > "$O/valu.txt" 2>&1
{ ./tools/ubench/conc; GPU_MAX_HW_QUEUES=8 ./tools/ubench/conc; GPU_MAX_HW_QUEUES=16 ./tools/ubench/conc; } > "$O/conc.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L > "$O/avail.txt" 2>&1)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "VALUBusy" "SALUBusy" "MemUnitBusy" "MemUnitStalled" "ALUStalledByLDS" "VALUUtilization" "TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum" "LdsUtil" "LdsLatency" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d "$O/pmc/pass$i" -- python "$REPO/bench.py" --steps 2 --warmup 1 --queue 8 --no-cpu --no-extras > "$O/pmc_pass$i.log" 2>&1
  echo "pass $i rc=$? : $grp" >> "$O/pmc_passes.txt"
done
cd "$REPO"
python tools/pmc_summary.py "$O/pmc" --json "$O/pmc_units.json" > "$O/pmc_units.txt" 2>&1
rm -rf "$O"/pmc/pass*/
echo done

#!/bin/bash
# round 5, second GPU probe: the VALU issue rate (clean loop) with its PMC counters; five wavefronts per SIMD on shorter sentences (= a smaller LDS footprint per sentence)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p2; mkdir -p "$O"
./tools/ubench/valu > "$O/valu.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$O/valu_pmc" -- "$REPO/tools/ubench/valu" > "$O/valu_pmc.log" 2>&1)
W5=$REPO/kanpyo_amd/libkanpyo_gpu_wpe5.so
c() { echo "== $*"; env "$@" BENCH_Q=8 python tools/bench_cfg.py cfg2 400000 4096 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for mean in 40 34 30 26; do
  c KANPYO_CFG2_MEAN=$mean KGPU_POOL=40:4:40
  c KANPYO_CFG2_MEAN=$mean KGPU_LIB=$W5 KGPU_POOL=32:4:40
  c KANPYO_CFG2_MEAN=$mean KGPU_LIB=$W5 KGPU_POOL=40:5:40
done
} > "$O/wpe5_by_length.txt" 2>&1
echo done

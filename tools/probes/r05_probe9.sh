#!/bin/bash
# round 5: the pair table with an odd row stride (-DKGPU_PAIR_ODD) against the shipped power-of-two-free plain stride: rates, LDS bank conflicts, parity
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p9; mkdir -p "$O"
ODD=$REPO/kanpyo_amd/libkanpyo_gpu_pairodd.so
KGPU_LIB=$ODD timeout 200 python tools/fuzz_parity.py 90 20261102 > "$O/fuzz_odd.txt" 2>&1; tail -1 "$O/fuzz_odd.txt"
bash tools/ab.sh -r 3 -c bench plain odd:KGPU_LIB=$ODD > "$O/ab_cfg2.txt" 2>&1
bash tools/ab.sh -r 2 -c window:cfg5 plain odd:KGPU_LIB=$ODD > "$O/ab_cfg5.txt" 2>&1
bash tools/ab.sh -r 2 -c cfg3:400000:4096 plain odd:KGPU_LIB=$ODD > "$O/ab_cfg3.txt" 2>&1
cat "$O"/ab_*.txt
cd /tmp && export TMPDIR=/tmp
for v in plain odd; do
  [ $v = odd ] && export KGPU_LIB=$ODD
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_$v/pass1" -- python "$REPO/bench.py" --steps 2 --warmup 1 --queue 1 --no-cpu --no-extras > "$O/pmc_$v.log" 2>&1
  BENCH_Q=8 timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_$v/pass2" -- python "$REPO/tools/bench_cfg.py" cfg5 1000 > "$O/pmc5_$v.log" 2>&1
  python "$REPO/tools/pmc_summary.py" "$O/pmc_$v" > "$O/pmc_$v.txt" 2>&1
  rm -rf "$O/pmc_$v"
done
grep -A6 "k_tokenize_pool<false, false> \[full\|k_tokenize_window<false, 1>" "$O"/pmc_plain.txt "$O"/pmc_odd.txt | grep -v "^--"

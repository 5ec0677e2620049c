#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/e2e_probe.txt; : > $OUT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) nproc: $(nproc) affinity: $(taskset -p $$ 2>/dev/null)" | tee -a $OUT
export KGPU_TEST_HOOKS_REREAD=1
run() { timeout 300 python tools/e2e_probe.py "$@" 2>&1 | grep -E "^e2e|kgpu_tokenize_batch:" | tail -3 | tee -a $OUT; }
KGPU_HOST_TRACE=1 run 4 pinned
KGPU_HOST_TRACE=1 run 4
KGPU_HOST_DEPTH=12 run 4 pinned
KGPU_HOST_DEPTH=16 run 4 pinned
KGPU_HOST_THREADS=16 run 4 pinned
KGPU_HOST_THREADS=4 run 4 pinned
KGPU_HOST_CHUNK_SENTS=8192 KGPU_HOST_DEPTH=12 run 4 pinned
KGPU_HOST_CHUNK_SENTS=4096 KGPU_HOST_DEPTH=16 run 4 pinned
KGPU_STREAMS=6 KGPU_HOST_DEPTH=12 run 4 pinned
KGPU_HOST_DEPTH=12 run 10 pinned
# long-sentence kernel with the text staged in LDS (A/B)
for lib in libkanpyo_gpu.so libkanpyo_gpu_ltext.so; do
  echo -n "$lib cfg3: " | tee -a $OUT; KGPU_LIB=$PWD/kanpyo_amd/$lib BENCH_Q=8 timeout 200 python tools/bench_cfg.py cfg3 400000 16384 2>&1 | tail -1 | tee -a $OUT
  echo -n "$lib cfg5: " | tee -a $OUT; KGPU_LIB=$PWD/kanpyo_amd/$lib BENCH_Q=8 timeout 200 python tools/bench_cfg.py cfg5 5000 1000 2>&1 | tail -1 | tee -a $OUT
done

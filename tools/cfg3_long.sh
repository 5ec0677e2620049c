export GPU_MAX_HW_QUEUES=8 BENCH_Q=8 KGPU_POOL=40:4:48
for l in 12 8 10 16 24; do
  echo -n "KGPU_LONG=$l : "; KGPU_LONG=$l timeout 150 python tools/bench_cfg.py cfg3 200000 2>&1 | tail -1
  echo -n "KGPU_LONG=$l : "; KGPU_LONG=$l timeout 150 python tools/bench_cfg.py cfg5 8000 1000 2>&1 | tail -1
done

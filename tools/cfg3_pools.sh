export GPU_MAX_HW_QUEUES=8 BENCH_Q=8
for pool in "40:4:48" "40:4:48,40:1:64" "40:4:48,80:2:64" "40:4:48,80:1:64" "40:4:40,53:1:64"; do
  echo -n "KGPU_POOL=$pool : "; KGPU_POOL=$pool timeout 150 python tools/bench_cfg.py cfg3 200000 2>&1 | tail -1
done

timeout 900 python -m pytest tests -m gpu -x -q --timeout 200 -k "launch_chain or cfg5 or edge" 2>&1 | tail -3
echo -n "cfg3: "; BENCH_Q=3 python tools/bench_cfg.py cfg3 20000 4096 2>&1 | tail -1
echo -n "cfg5: "; BENCH_Q=2 python tools/bench_cfg.py cfg5 1000 1000 2>&1 | tail -1
for k in 3 5 7; do echo -n "cfg5 stop $k: "; KGPU_POOL=0 KGPU_DEBUG_STOP=$k BENCH_Q=2 python tools/bench_cfg.py cfg5 1000 1000 2>&1 | tail -1 | sed 's/.*(\(.*\))/\1/'; done
for k in 3 5 7; do echo -n "cfg3 stop $k: "; KGPU_DEBUG_STOP=$k BENCH_Q=3 python tools/bench_cfg.py cfg3 20000 4096 2>&1 | tail -1 | sed 's/.*(\(.*\))/\1/'; done

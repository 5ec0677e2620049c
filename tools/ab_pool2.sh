#!/bin/bash
# two-pool chains (KGPU_POOL holds a comma: tools/ab.sh splits its settings on commas)
run() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu --no-extras --no-stages 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(round(d['value']/1e6,2), 'M sentences/s, kernel ms in flight / alone', r.get('avg_kernel_ms'), r.get('kernel_alone_ms'))"; python -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print('   ', d['routing'])"; }
for r in 1 2; do
  for p in "$@"; do echo "[$p] $(KGPU_POOL=$p run)"; done
done

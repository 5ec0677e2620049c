#!/bin/bash
# round 5: the host-side expansion of the 8-byte records with non-temporal stores (KGPU_EXPAND_STREAM=0: ordinary stores as before)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p23; mkdir -p "$O"
{
for r in 1 2; do for m in 0 1; do
  echo "== KGPU_EXPAND_STREAM=$m"; KGPU_EXPAND_STREAM=$m timeout 300 python tools/e2e_quick.py 2>&1 | grep -v amdgpu.ids | tail -2
  KGPU_EXPAND_STREAM=$m python -c "
from kanpyo_amd.tokenizer import merge_bench
r = merge_bench(8, 8192, 32, reps=20); print('merge alone: %.1f M sentences/s' % (r['sentences_per_s'] / 1e6))"
done; done
} > "$O/expand_stream.txt" 2>&1
cat "$O/expand_stream.txt"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -2

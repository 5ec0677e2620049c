#!/bin/bash
# round 5: the pool kernel's sentences handed to a workgroup's wavefronts by an LDS ticket (shipped build) against the static every-W-th form (-DKGPU_POOL_STATIC)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p20; mkdir -p "$O"
ST=$REPO/kanpyo_amd/libkanpyo_gpu_static.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_concurrent.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab.sh -r 3 -c bench ticket static:KGPU_LIB=$ST > "$O/ab_cfg2.txt" 2>&1; cat "$O/ab_cfg2.txt"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for lib in "" "KGPU_LIB=$ST"; do
  c BENCH_Q=1 $lib A=1 python tools/bench_cfg.py cfg2 400000 4096
  c BENCH_Q=1 $lib A=1 python tools/bench_cfg.py cfg2 400000 16384
  c BENCH_Q=1 $lib A=1 python tools/bench_cfg.py cfg2 400000 65536
  c BENCH_Q=8 $lib A=1 python tools/bench_cfg.py cfg3 400000 65536
  c BENCH_Q=8 $lib A=1 python tools/bench_cfg.py cfg3 400000 4096
  c BENCH_Q=8 $lib A=1 python tools/bench_cfg.py dense 200000 4096
done
} > "$O/others.txt" 2>&1
cat "$O/others.txt"

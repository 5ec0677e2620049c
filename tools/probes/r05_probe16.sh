#!/bin/bash
# round 5: scan + compaction of pool-only chains on a partner stream of each shared stream (KGPU_AUX_STREAMS=1): does the shared stream's next pool launch start sooner?
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p16; mkdir -p "$O"
KGPU_AUX_STREAMS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab.sh -r 3 -c bench plain aux:KGPU_AUX_STREAMS=1 > "$O/ab_cfg2.txt" 2>&1
cat "$O/ab_cfg2.txt"
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
c BENCH_Q=8 KGPU_AUX_STREAMS=1 python tools/bench_cfg.py dense 200000 4096
c BENCH_Q=8 python tools/bench_cfg.py dense 200000 4096
c BENCH_Q=1 KGPU_AUX_STREAMS=1 python tools/bench_cfg.py cfg2 400000 4096
c BENCH_Q=1 python tools/bench_cfg.py cfg2 400000 4096
} > "$O/others.txt" 2>&1
cat "$O/others.txt"

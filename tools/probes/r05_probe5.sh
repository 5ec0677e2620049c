#!/bin/bash
# round 5, fifth GPU probe: does asking for 16 hardware queues cost cfg 2 anything; cfg 3 with more queues / streams
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p5; mkdir -p "$O"
b() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2), 'M sentences/s, streams', d['config']['streams'], 'alone ms', round(d['roofline'].get('kernel_alone_ms',0),4), 'kernel ms', round(d['roofline']['avg_kernel_ms'],4))"; }
{
b A=1
b GPU_MAX_HW_QUEUES=16
b GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=5
b GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=6
b GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8
b GPU_MAX_HW_QUEUES=12
b A=2
} > "$O/cfg2_queues.txt" 2>&1
c() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 GPU_MAX_HW_QUEUES=16 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=16 GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=12 GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=6 python tools/bench_cfg.py cfg3 400000 4096
c BENCH_Q=8 python tools/bench_cfg.py cfg3 400000 65536
c BENCH_Q=8 GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 python tools/bench_cfg.py cfg3 400000 65536
} > "$O/cfg3_queues.txt" 2>&1
echo done

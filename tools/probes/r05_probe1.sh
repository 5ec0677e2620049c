#!/bin/bash
# round 5, first GPU probe: VALU issue rate (ubench + PMC), streams x contexts on cfg 5 / cfg 3, five wavefronts per SIMD for the pool kernel
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p1; mkdir -p "$O"
./tools/ubench/valu > "$O/valu.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/valu_pmc" -- "$REPO/tools/ubench/valu" > "$O/valu_pmc.log" 2>&1)
w() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids; }
{
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 1000 8
w KGPU_STREAMS=8 python tools/window_timing.py cfg5 1000 8
w KGPU_STREAMS=8 python tools/window_timing.py cfg5 1000 16
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 1000 16
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 1000 1
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 1000 2
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 1000 4
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 4000 4
} > "$O/cfg5_streams.txt" 2>&1
{
w BENCH_Q=8 KGPU_STREAMS=4 python tools/bench_cfg.py cfg3 200000 4096
w BENCH_Q=16 KGPU_STREAMS=4 python tools/bench_cfg.py cfg3 200000 4096
w BENCH_Q=8 KGPU_STREAMS=4 python tools/bench_cfg.py cfg3 200000 65536
} > "$O/cfg3.txt" 2>&1
b() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2), 'M sentences/s, redone', d['routing']['redone'][0], 'alone ms', round(d['roofline'].get('kernel_alone_ms',0),4), 'kernel ms', round(d['roofline']['avg_kernel_ms'],4))"; }
{
b A=1
b KGPU_LIB=$REPO/kanpyo_amd/libkanpyo_gpu_wpe5.so KGPU_POOL=32:4:40
b KGPU_LIB=$REPO/kanpyo_amd/libkanpyo_gpu_wpe5.so KGPU_POOL=40:5:40
b KGPU_LIB=$REPO/kanpyo_amd/libkanpyo_gpu_wpe5.so KGPU_POOL=40:4:40
b KGPU_LIB=$REPO/kanpyo_amd/libkanpyo_gpu_wpe5.so KGPU_POOL=80:10:64
b A=2
} > "$O/pool_wpe5.txt" 2>&1
echo done

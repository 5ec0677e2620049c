#!/bin/bash
# ONE A/B driver for the GPU box (replaces the ~30 one-off ab_*.sh / cfg3_*.sh / window_*.sh / pool_*.sh scripts of rounds 2-3).
#
#   bash tools/ab.sh [-r REPS] [-c COMMAND] VARIANT...
#
# A VARIANT is "label" or "label:ENV=val,ENV2=val,..." (a ';' inside a value stands for a comma: KGPU_POOL=10:1:64;40:4:32) -- environment settings for one arm (KGPU_LIB=<other build of the library>, KGPU_POOL=40:4:48,
# KGPU_WINDOW=12, KGPU_STREAMS=..., GPU_MAX_HW_QUEUES=..., BENCH_Q=...).  The arms run interleaved, REPS times each (default 2), on the same box.
# COMMAND (default: the bench.py headline) is one of
#   bench                 python bench.py --no-cpu --no-extras --no-stages  -> M sentences/s (value), the pool kernel's launch duration in flight / alone
#   dense[:n[:batch]]     python tools/bench_cfg.py dense n batch           -> cfg 2-shaped text over the dense-lattice dictionary
#   cfg3[:n[:batch]]      python tools/bench_cfg.py cfg3 n batch            -> the tool's result line       (defaults 400000, 65536)
#   cfg5[:n[:batch]]      python tools/bench_cfg.py cfg5 n batch                                            (defaults 1000, 4096)
#   window:cfg5|cfg3      python tools/window_timing.py ...                 -> rate + shader clocks per character by phase of the windowed kernel
#   callers[:threads]     python tools/callers_probe.py threads 200 2       -> concurrent one-sentence callers (with the cgroup's CPU accounting)
#   onectx                python tools/one_ctx_probe.py                     -> one / two contexts on cfg 2, host-call latencies at n = 1, 64, 4096
#   anything else         run as given (quote it)
# Example (round 4, pool routing limit x windowed-kernel LDS on cfg 3):
#   bash tools/ab.sh -c cfg3 p40w12:KGPU_POOL=40:4:40,KGPU_WINDOW=12 p48w12:KGPU_POOL=40:4:48,KGPU_WINDOW=12 p48w16:KGPU_POOL=40:4:48,KGPU_WINDOW=16
REPS=2; CMD=bench
while getopts "r:c:" o; do case $o in r) REPS=$OPTARG;; c) CMD=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
cd "$(dirname "$0")/.."
IFS=: read -r kind a1 a2 <<< "$CMD"
case $kind in
  bench)   run() { timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu --no-extras --no-stages 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(round(d['value']/1e6,2), 'M sentences/s, kernel ms in flight / alone', r.get('avg_kernel_ms'), r.get('kernel_alone_ms'))"; } ;;
  dense)   run() { timeout 300 python tools/bench_cfg.py dense 100000 4096 2>&1 | grep -v amdgpu.ids | tail -1; } ;;
  cfg3)    run() { timeout 600 python tools/bench_cfg.py cfg3 ${a1:-400000} ${a2:-65536} 2>&1 | grep -v amdgpu.ids | tail -1; } ;;
  cfg5)    run() { timeout 600 python tools/bench_cfg.py cfg5 ${a1:-1000} ${a2:-4096} 2>&1 | grep -v amdgpu.ids | tail -1; } ;;
  window)  run() { timeout 600 python tools/window_timing.py ${a1:-cfg5} $([ "${a1:-cfg5}" = cfg3 ] && echo 60000 || echo 1000) 8 2>&1 | grep -v amdgpu.ids | head -2; } ;;
  callers) run() { timeout 600 python tools/callers_probe.py ${a1:-64} 200 2 2>&1 | grep -v amdgpu.ids; } ;;
  onectx)  run() { timeout 600 python tools/one_ctx_probe.py 2>&1 | grep -v amdgpu.ids; } ;;
  *)       run() { timeout 900 bash -c "$CMD" 2>&1 | grep -v amdgpu.ids | tail -3; } ;;
esac
export BENCH_Q=${BENCH_Q:-8}
for r in $(seq "$REPS"); do
  for v in "$@"; do
    label=${v%%:*}; envs=""; [ "$v" != "$label" ] && envs=${v#*:}
    echo "[$label] $(env $(echo "$envs" | tr ',' ' ' | tr ';' ',') bash -c "$(declare -f run); kind=$kind a1=$a1 a2=$a2 CMD='$CMD' run")"
  done
done

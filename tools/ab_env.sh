#!/bin/bash
# tools/ab_env.sh -- A/B of environment settings on one box, interleaved: every arm's headline (bench.py --no-extras --no-cpu --no-stages) REPS times.
# usage (one gpurun call): bash tools/ab_env.sh OUT.txt REPS "NAME1:VAR=V VAR2=V" "NAME2:" ...   (an arm with nothing behind the colon is the baseline)
out=$1; reps=$2; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq 1 "$reps"); do
  for arm in "$@"; do
    name=${arm%%:*}; envs=${arm#*:}
    v=$(env $envs python bench.py --steps 40 --warmup 10 --no-extras --no-cpu --no-stages 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), r.get('avg_kernel_ms'), r.get('kernel_alone_ms'))")
    echo "$name rep$r $v" | tee -a "$out"
  done
done

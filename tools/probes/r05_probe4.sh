#!/bin/bash
# round 5, fourth GPU probe: cfg 5 with more hardware queues / streams, with the pool (router) launch left out, and a kernel timeline of the 8-stream run
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd "$REPO"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth
O=$REPO/gpurun_out/p4; mkdir -p "$O"
w() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | head -2; }
{
w KGPU_STREAMS=4 python tools/window_timing.py cfg5 1000 8
w KGPU_STREAMS=4 KGPU_POOL=0 python tools/window_timing.py cfg5 1000 8
w GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 python tools/window_timing.py cfg5 1000 8
w GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 KGPU_POOL=0 python tools/window_timing.py cfg5 1000 8
w GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 KGPU_POOL=0 python tools/window_timing.py cfg5 1000 16
w GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=12 KGPU_POOL=0 python tools/window_timing.py cfg5 1000 12
w GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=6 KGPU_POOL=0 python tools/window_timing.py cfg5 1000 12
w KGPU_STREAMS=4 KGPU_POOL=0 python tools/window_timing.py cfg5 1000 1
w KGPU_STREAMS=4 KGPU_POOL=0 python tools/window_timing.py cfg5 4000 4
} > "$O/cfg5.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && GPU_MAX_HW_QUEUES=16 KGPU_STREAMS=8 KGPU_POOL=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace8" -- python "$REPO/tools/window_timing.py" cfg5 1000 8 > "$O/trace8.log" 2>&1)
(cd /tmp && export TMPDIR=/tmp && KGPU_STREAMS=4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace4" -- python "$REPO/tools/window_timing.py" cfg5 1000 8 > "$O/trace4.log" 2>&1)
python - <<'PY' > "$O/trace_summary.txt" 2>&1
import csv, glob, collections
for tag in ("trace8", "trace4"):
    fs = glob.glob(f"gpurun_out/p4/{tag}/**/*kernel_trace.csv", recursive=True)
    if not fs: print(tag, "no trace"); continue
    rows = list(csv.DictReader(open(fs[0])))
    rows = [r for r in rows if "k_tokenize_window<false>" in r["Kernel_Name"]]
    ev = []
    for r in rows: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    hist = collections.Counter(); cur = 0; last = ev[0][0]
    for t, d in ev:
        hist[cur] += t - last; last = t; cur += d
    tot = sum(hist.values())
    durs = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    print(tag, "window launches", len(rows), "median ms", durs[len(durs)//2] / 1e6, "p10", durs[len(durs)//10] / 1e6, "p90", durs[9*len(durs)//10] / 1e6)
    print("   time share by number of window kernels running side by side:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
    queues = collections.Counter(r["Queue_Id"] for r in rows)
    print("   launches per queue:", dict(queues))
PY
rm -rf "$O/trace8" "$O/trace4"
echo done

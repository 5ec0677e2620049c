#!/bin/bash
# GPU box: everything profiles/ holds for a round -- the bench line, rocprofv3 kernel-trace stats of the same command, the PMC
# passes (pool kernel: cfg 2; long-sentence kernel: cfg 5 and cfg 3), per-phase counters.
# usage: bash tools/collect_profiles.sh <outdir> <round tag, e.g. r02> [pool|headline]      ("pool": the cfg 2 / pool-kernel part only -- about a third of the time;
#        "headline": the bench line and the kernel trace of the same command only -- when bench.py changed and the library did not: ~4 minutes)
set -u
OUT=$(realpath -m "$1"); TAG=$2; PART=${3:-all}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export KANPYO_SYNTH_CACHE=/tmp/kanpyo_synth   # (the synthetic dictionary is built once per box, not once per process)
# the plain bench line: no GPU_MAX_HW_QUEUES in the environment -- the library sets it itself (config.streams says whether that took effect)
# (bench.py prints the compact line and writes the full report to bench_full.json beside itself: both are kept)
(unset GPU_MAX_HW_QUEUES; python "$REPO/bench.py" > "$OUT/${TAG}_bench_line.json" 2> "$OUT/bench.err"; cp "$REPO/bench_full.json" "$OUT/${TAG}_bench.json")
# under rocprofv3 the profiler initialises the runtime before the library is loaded: there the variable comes from the environment
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu --no-extras --no-stages > "$OUT/trace_line.json" 2> "$OUT/trace.err"
cp "$REPO/bench_full.json" "$OUT/trace.json"
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_kernel_stats.csv"
# the same trace per dispatch: k_tokenize_pool by grid size (full 4096-sentence batches, the ragged last batch of a pass, small calls)
{ echo "# rocprofv3 --kernel-trace of: python bench.py --steps 20 --warmup 5 --no-cpu --no-extras; bench line of the same run: trace.json"; python "$REPO/tools/trace_pool.py" "$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)" 480; python - "$OUT/trace.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"bench.py (same run, HIP events): value {d['value']:.0f} sentences/s, avg_kernel_ms {r['avg_kernel_ms']:.4f}, avg_launch_chain_ms {r['avg_launch_chain_ms']:.4f}, launches_timed {r['launches_timed']}")
PY
} > "$OUT/${TAG}_pool_dispatches.txt"
if [ "$PART" = headline ]; then rm -rf "$OUT"/trace; echo done; exit 0; fi
if [ "$PART" = all ]; then
# cfg 5 through tools/window_timing.py (one 1000-document batch per context): eight contexts = eight launches in flight = the ordinary, one-wavefront-per-document
# form of the windowed kernel (k_tokenize_window<false, 1>); one context = a lone batch = its team form (<false, 2>).  cfg 3 through tools/bench_cfg.py.
for q in 8 1; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg5_q$q" -- python "$REPO/tools/window_timing.py" cfg5 1000 $q > "$OUT/trace_cfg5_q$q.log" 2>&1
  cp "$(find "$OUT/trace_cfg5_q$q" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_cfg5_q${q}_kernel_stats.csv"
done
BENCH_Q=8 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg3" -- python "$REPO/tools/bench_cfg.py" cfg3 100000 > "$OUT/trace_cfg3.log" 2>&1
cp "$(find "$OUT/trace_cfg3" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_cfg3_kernel_stats.csv"
fi
bash "$REPO/tools/pmc_passes.sh" "$OUT/pmc" python "$REPO/bench.py" --steps 2 --warmup 1 --queue 1 --no-cpu --no-extras --no-stages > "$OUT/pmc.log" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT/pmc" --json "$OUT/${TAG}_pmc_summary.json" > "$OUT/${TAG}_pmc_summary.txt"
python "$REPO/tools/make_traffic_json.py" "$OUT/${TAG}_pmc_summary.json" "$OUT/pmc_traffic.json" "$TAG" > /dev/null
if [ "$PART" = all ]; then
# cfg 5: eight contexts = the ordinary (one wavefront per document) form of the windowed kernel; one context = its team form (two wavefronts per document)
bash "$REPO/tools/pmc_passes.sh" "$OUT/pmc_cfg5" python "$REPO/tools/window_timing.py" cfg5 1000 8 > "$OUT/pmc_cfg5.log" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT/pmc_cfg5" --json "$OUT/${TAG}_cfg5_pmc_summary.json" > "$OUT/${TAG}_cfg5_pmc_summary.txt"
bash "$REPO/tools/pmc_passes.sh" "$OUT/pmc_cfg5_team" python "$REPO/tools/window_timing.py" cfg5 1000 1 > "$OUT/pmc_cfg5_team.log" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT/pmc_cfg5_team" --json "$OUT/${TAG}_cfg5_team_pmc_summary.json" > "$OUT/${TAG}_cfg5_team_pmc_summary.txt"
fi
bash "$REPO/tools/pmc_phases.sh" "$OUT/phases" 1 2 3 4 5 6 7 0 > "$OUT/${TAG}_phase_counters.txt" 2>&1
# the bench line once more, now that the counter files belong to this tree (traffic_stale false)
cp "$OUT/pmc_traffic.json" "$OUT/pmc_instructions.json" "$REPO/profiles/"
cp "$OUT/${TAG}_bench.json" "$OUT/${TAG}_bench_first.json"
(cd "$REPO" && unset GPU_MAX_HW_QUEUES && python bench.py > "$OUT/${TAG}_bench_line.json" 2> "$OUT/bench2.err"; cp "$REPO/bench_full.json" "$OUT/${TAG}_bench.json")
(cd "$REPO/kanpyo_amd/csrc" && make -s resource-usage 2>&1 | grep -E "Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|Occupancy|LDS Size" > "$OUT/${TAG}_resource_usage.txt")
rm -rf "$OUT"/trace "$OUT"/trace_cfg5_q8 "$OUT"/trace_cfg5_q1 "$OUT"/trace_cfg3 "$OUT"/pmc/pass*/ "$OUT"/pmc_cfg5/pass*/ "$OUT"/pmc_cfg5_team/pass*/ "$OUT"/phases
echo done

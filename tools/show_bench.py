#!/usr/bin/env python
"""Print the headline fields of a bench.py JSON line: python tools/show_bench.py gpurun_out/x/bench.json"""
import json, sys
p = json.load(open(sys.argv[1]))
r = p["roofline"]
print(f"value {p['value']/1e6:.2f} M sentences/s, ms/step {p['ms_per_step']:.3f}, e2e {p.get('value_end_to_end',{}).get('value',0)/1e6:.2f} M")
print(f"roofline: per launch {r['frac']:.4f} ({r['avg_kernel_ms']*1e3:.1f} us), alone {r.get('frac_alone')} ({r.get('kernel_alone_ms')}), job rate {r['frac_at_job_rate']:.4f}")
if "stages" in r:
    print("stages:", {k: (round(v["ms_per_step"], 3), round(v["frac"], 3) if v["frac"] is not None else None) for k, v in r["stages"].items() if isinstance(v, dict) and "ms_per_step" in v})
print("routing:", p["routing"])
pi = p.get("pcie_inclusive", {})
print("latency:", {k: round(v["median_us"], 1) for k, v in pi.get("call_latency", {}).items()}, "4096/call", round(pi.get("value", 0) / 1e6, 2), "M")
for k, v in pi.get("concurrent_callers", {}).items():
    if isinstance(v, dict):
        print(f"  {k}: {v['sentences_per_s']/1e3:.0f} k sentences/s, p50 {v['p50_us']:.0f} us, p99 {v['p99_us']:.0f}, per launch {v['sentences_per_launch']:.1f}, combined {v['combined_calls']}/{v['small_calls']}")
cb = p.get("cpu_baseline")
if cb:
    print(f"cpu 1 thread {cb['value']/1e3:.1f} k, all {cb['all_cores']['value']/1e6:.2f} M on {cb['all_cores']['cores']}, exact {cb['gpu_batch0_bit_exact']}")
for e in p.get("extra", []):
    print(f"extra: {e['workload'][:70]}...: {e['value']/1e6:.3f} M sentences/s, {e['Mchar_per_s']:.0f} Mchar/s, batch {e['batch']}, exact {e['first_batch_bit_exact_vs_oracle']}, routing {e['routing']['deferred']} redone {e['routing']['redone']}")
    if "lattice_density" in e:
        print("   ", e["lattice_density"], "relax/s vs cfg2", round(e["relaxations_per_s_vs_cfg2"], 3), {k: round(v, 1) for k, v in e["work_per_sentence"].items()})

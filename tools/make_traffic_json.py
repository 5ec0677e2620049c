#!/usr/bin/env python
"""profiles/pmc_traffic.json from a pmc_summary.json (tools/pmc_summary.py --json): HBM bytes per
batch of the tokenize kernels, FETCH_SIZE corrected as MI355X_MICROARCH.md's HBM section prescribes.
Also writes pmc_instructions.json next to it (wave-instructions per sentence of the pool kernel: bench.py's instruction roofline).
usage: python tools/make_traffic_json.py <pmc_summary.json> <out.json> [round tag]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kanpyo_amd._lib import kernel_source_hash  # the sources the profiled library was built from (run this on the tree that was profiled)

src, dst = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
d = json.load(open(src))
# the product instantiation only (k_tokenize_pool<false, ...>): the profiling one (<true, ...>: device-side work counters, and the byte-level
# walk beside the product's) runs outside the timed region
FULL = " [full 4096-sentence launches]"
kernels = [k for k in d if "k_tokenize_pool" in k and "pool<true" not in k and k.endswith(FULL)]
main = kernels[0]
batches = d[main]["FETCH_SIZE"]["dispatches"]
fetch_kb = d[main]["FETCH_SIZE"]["per_dispatch"]  # per launch of the dominant kernel over a full batch (the general kernel
write_kb = d[main]["WRITE_SIZE"]["per_dispatch"]  # behind it finds an empty list on cfg 2)
out = {
    "kernel_src_sha16": kernel_source_hash(),
    "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
              "`python bench.py --steps 2 --warmup 1 --queue 1 --no-cpu --no-extras`, tools/pmc_passes.sh)",
    "per": "one k_tokenize_pool launch over a full batch of 4096 sentences (grid 1024 x 256): " + main,
    "batches": batches,
    "fetch_size_kb": fetch_kb,
    "write_size_kb": write_kb,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128 B request on gfx950 -> x2 (calibrated "
                  "for wide coalesced reads; this kernel issues narrow gathers, so x2 is an upper bound); WRITE_SIZE "
                  "uncalibrated, taken as is; unit KB = 1024 B",
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "hbm_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

m = d[main]
waves = m["SQ_WAVES"]["sum"]  # one wavefront per sentence (4096-sentence batches, 4096 wavefronts)
ins = {
    "kernel_src_sha16": kernel_source_hash(),
    "source": f"profiles/{tag}_pmc_summary.json, kernel {main}: SQ_INSTS_* / SQ_WAVES (every wavefront of a full batch tokenizes one sentence)",
    "valu_per_sentence": m["SQ_INSTS_VALU"]["sum"] / waves, "salu_per_sentence": m["SQ_INSTS_SALU"]["sum"] / waves,
    "lds_per_sentence": m["SQ_INSTS_LDS"]["sum"] / waves, "vmem_rd_per_sentence": m["SQ_INSTS_VMEM_RD"]["sum"] / waves,
    "wave_cycles_per_sentence": 4 * m["SQ_WAVE_CYCLES"]["sum"] / waves,
}
json.dump(ins, open(os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_instructions.json"), "w"), indent=1)

import sys, json
sys.path.insert(0, '.')
from kanpyo_amd.tokenizer import merge_bench
for c in (False, True):
    r = merge_bench(8, 8192, 32, reps=20, compact=c)
    print("compact" if c else "24-byte", round(r["sentences_per_s"] / 1e6, 1), "M sentences/s", round(r["bytes_moved_GB_per_s"], 1), "GB/s")

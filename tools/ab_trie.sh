#!/bin/bash
# character-level trie against the byte-level walk (KGPU_BYTE_TRIE=1: no char-level copy), same library, interleaved -> gpurun_out/ab_trie.txt
mkdir -p gpurun_out; OUT=gpurun_out/ab_trie.txt; : > $OUT
for r in $(seq ${1:-3}); do for bt in 0 1; do
  v=$(KGPU_BYTE_TRIE=$bt timeout 200 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['value']/1e6,2))")
  echo "KGPU_BYTE_TRIE=$bt $v" | tee -a $OUT
done; done

#!/bin/bash
# GPU box, end of round 4: the gpu test suite, a fuzz run, the PMC passes on the final tree (counter files carry its hash), then the bench line.
cd /root/repo; OUT=gpurun_out/final_r04; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $OUT/tests.log
timeout 500 python tools/fuzz_parity.py 330 20261002 2>&1 | tail -2 > $OUT/fuzz.log
bash tools/collect_pmc.sh $OUT r04 > $OUT/pmc_collect.log 2>&1
cp $OUT/pmc_traffic.json $OUT/pmc_instructions.json profiles/
cp $OUT/r04_pmc_summary.txt $OUT/r04_pmc_summary.json profiles/
(unset GPU_MAX_HW_QUEUES; python bench.py > $OUT/r04_bench.json 2> $OUT/bench.err)
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
cat $OUT/tests.log $OUT/fuzz.log $OUT/smoke.log; python tools/show_bench.py $OUT/r04_bench.json | head -8

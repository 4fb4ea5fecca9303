#!/bin/bash
# round 4, first GPU call: multiply micro-benchmark, the finish-kernel LDS experiment, baseline bench
set -x
O=gpurun_out/r04_first; mkdir -p $O
./tools/ubench/mulround > $O/mulround.jsonl 2>&1
timeout 1500 python tools/fin_lds_experiment.py run 6 > $O/fin_lds.jsonl 2> $O/fin_lds.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
cat $O/mulround.jsonl $O/fin_lds.jsonl

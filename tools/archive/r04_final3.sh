#!/bin/bash
# Round 4 at HEAD, last run: the whole GPU tier, smoke(), the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_final3}
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.err
( timeout 100 python tests/stress/stress_raht_inter_gpu.py 7750000 60 ) > $OUT/stress_raht_inter_gpu.txt 2>&1; tail -n 2 $OUT/stress_raht_inter_gpu.txt

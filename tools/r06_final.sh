#!/bin/bash
# Round 6 at HEAD: the whole GPU tier, smoke(), the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06_final}
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -n 6 $OUT/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -n 3 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.err
head -c 600 $OUT/bench_default.json

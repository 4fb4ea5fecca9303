#!/bin/bash
# the whole GPU tier + the default bench line (+ its rocprofv3 kernel trace and PMC passes) at HEAD
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_full}
mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.err

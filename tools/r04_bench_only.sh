#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_bench_only}; mkdir -p $O
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 200 $O/bench_default.err
timeout 300 python -m pytest tests/test_bench_multirank.py -m gpu -q > $O/multirank.log 2>&1; tail -n 2 $O/multirank.log

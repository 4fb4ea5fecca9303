#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_inter_quick}; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_raht_inter.py -m gpu -q ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -n 25 $O/pytest.log

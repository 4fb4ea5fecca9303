#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_tier}; mkdir -p $O
( time AMD_LOG_LEVEL=1 timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^:1:\|hip_" $O/pytest.log | tail -n 6

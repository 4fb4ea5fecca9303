#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_half}; mkdir -p $O
( time AMD_LOG_LEVEL=1 timeout 135 python -m pytest tests/test_gpu_[m-z]*.py tests/test_shim*.py tests/test_zz*.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "rc $?"
grep -v "^:1:\|hip_" $O/pytest.log | tail -n 5 | cut -c1-160

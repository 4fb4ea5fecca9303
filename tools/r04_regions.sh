#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_regions}; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_regions.py tests/test_shim_operator.py tests/test_gpu_lift.py tests/test_gpu_pred.py -m gpu -q ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -n 8 $O/pytest.log

#!/bin/bash
# register floor of the finish kernel (pinned batches + random ragged batches), both arithmetic back ends,
# the operator with RAHT slices through seam 3 and two-attribute slices, operator-level wall time
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_ops}; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_batches.py tests/test_gpu_arith.py tests/test_shim_operator.py tests/test_shim_dropin.py -m gpu -q ) > $O/pytest.log 2>&1; tail -n 12 $O/pytest.log
timeout 600 python tools/operator_time.py > $O/operator_time.txt 2>&1; tail -n 12 $O/operator_time.txt

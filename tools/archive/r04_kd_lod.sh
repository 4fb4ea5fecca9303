#!/bin/bash
# kd-tree build with chunked bounds / counts and 1 024-point subtrees; LoD search in two launches
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_kd_lod}; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_recolour.py tests/test_gpu_lod.py tests/test_gpu_lift.py tests/test_gpu_pred.py tests/test_zz_gpu_inter_lod.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -n 6 $O/pytest.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/ktrc -o kt -- bash -c "cd $GRAFT_REPO_ROOT && python tools/recolour_time.py" > $GRAFT_REPO_ROOT/$O/ktrc.log 2>&1 )
tail -n 3 $O/ktrc.log
head -14 $(find $O/ktrc -name '*kernel_stats.csv' | head -1) | cut -c1-150
timeout 600 python tools/lift_time.py > $O/lift_time.txt 2>&1; tail -n 12 $O/lift_time.txt
timeout 300 python tools/recolour_time.py > $O/recolour_time.txt 2>&1; tail -n 3 $O/recolour_time.txt

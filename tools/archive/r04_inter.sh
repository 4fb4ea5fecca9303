#!/bin/bash
# inter-frame RAHT on the device: parity tests and timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_inter}; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_raht_inter.py tests/test_gpu_recolour.py -m gpu -q ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -n 25 $O/pytest.log
timeout 600 python tools/raht_inter_time.py > $O/inter_time.txt 2>&1; tail -n 30 $O/inter_time.txt
GPCC_F64=0 GPCC_INTER_REF=0 timeout 600 python tools/raht_inter_time.py > $O/inter_time_int64.txt 2>&1; grep "subnode 1 decision 1" $O/inter_time_int64.txt | cut -c1-200

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_inter_quick}; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_raht_inter.py -m gpu -q ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -n 8 $O/pytest.log
GPCC_INTER_REF=0 timeout 600 python tools/raht_inter_time.py > $O/inter_time.txt 2>&1; grep "decision 1 estimated_taps 0" $O/inter_time.txt | sed 's/"modes.*"forward_kernels_ms"/"forward_kernels_ms"/' | cut -c1-420

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r04_subprof}; mkdir -p $O
for q in 34 22; do
  QP=$q GPCC_LIB_PATH=$PWD/exp/libgpcc_subprof.so timeout 300 python tools/sub_prof.py 1 forward > $O/subprof_fwd_qp$q.txt 2>&1; tail -n 16 $O/subprof_fwd_qp$q.txt
done

#!/bin/bash
O=gpurun_out/${1:-r04_prof}; mkdir -p $O
for a in 1 0; do
GPCC_F64=$a GPCC_LIB_PATH=$PWD/exp/libgpcc_subprof.so timeout 300 python tools/sub_prof.py 1 forward > $O/subprof_fwd_f64_$a.txt 2>&1; tail -n 14 $O/subprof_fwd_f64_$a.txt
GPCC_F64=$a GPCC_PIPE=0 GPCC_LIB_PATH=$PWD/exp/libgpcc_subprof.so timeout 300 python tools/sub_prof.py 1 inverse > $O/subprof_inv_f64_$a.txt 2>&1; tail -n 14 $O/subprof_inv_f64_$a.txt
done

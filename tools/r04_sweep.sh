#!/bin/bash
# headline of experiment builds exp/libgpcc_<name>.so (level-by-level decoder, ArithF64)
O=gpurun_out/${1:-r04_sweep}; mkdir -p $O; shift
for n in base "$@"; do
  lib=$PWD/exp/libgpcc_$n.so; [ $n = base ] && lib=$PWD/mpeg-pcc-tmc13_amd/libgpcc_attr_mi355.so
  GPCC_LIB_PATH=$lib GPCC_PIPE=${PIPE:-0} timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('$n', 'ms_per_step', d['ms_per_step'], 'fwd', {k:round(v,3) for k,v in r['forward_kernel_ms'].items() if v>0.3}, 'inv', {k:round(v,3) for k,v in r['inverse_kernel_ms'].items() if v>0.3})
except Exception as e:
    print('$n', 'ERR', e, open('$O/bench_$n.err').read()[-300:])
PY
done

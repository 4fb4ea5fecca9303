#!/bin/bash
# round 4: ArithF64 in the level-by-level sub-node kernels -- parity subset, headline A/B, stage profile of the old kernel
O=gpurun_out/${1:-r04_f64}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_pipe.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "f64_pipe:GPCC_F64=1" "i64_pipe:GPCC_F64=0" "f64_lvl:GPCC_F64=1 GPCC_PIPE=0" "i64_lvl:GPCC_F64=0 GPCC_PIPE=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('$name', 'ms_per_step', d['ms_per_step'], 'fwd', {k:round(v,3) for k,v in r['forward_kernel_ms'].items() if v>0.1}, 'inv', {k:round(v,3) for k,v in r['inverse_kernel_ms'].items() if v>0.1})
except Exception as e:
    print('$name', 'ERR', e, open('$O/bench_$name.err').read()[-300:])
PY
done
if [ -f exp/libgpcc_subprof.so ]; then
  GPCC_LIB_PATH=$PWD/exp/libgpcc_subprof.so timeout 300 python tools/sub_prof.py 1 forward > $O/subprof_fwd.txt 2>&1; cat $O/subprof_fwd.txt | tail -16
  GPCC_PIPE=0 GPCC_LIB_PATH=$PWD/exp/libgpcc_subprof.so timeout 300 python tools/sub_prof.py 1 inverse > $O/subprof_inv.txt 2>&1; cat $O/subprof_inv.txt | tail -16
fi

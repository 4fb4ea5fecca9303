#!/bin/bash
# quick GPU check of a change to the sub-node kernels: parity subset, headline in four configurations
O=gpurun_out/${1:-r04_quick}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_pipe.py tests/test_gpu_batches.py -m gpu -x -q > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
for cfg in ${CFGS:-"f64_lvl:GPCC_F64=1,GPCC_PIPE=0" "f64_pipe:GPCC_F64=1" "i64_lvl:GPCC_F64=0,GPCC_PIPE=0"}; do
  name=${cfg%%:*}; envs=$(echo ${cfg#*:} | tr ',' ' ')
  env $envs GPCC_PROFILE_LEVELS=${LEVELS:-0} timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('$name', 'ms_per_step', d['ms_per_step'], 'fwd', {k:round(v,3) for k,v in r['forward_kernel_ms'].items() if v>0.08}, 'inv', {k:round(v,3) for k,v in r['inverse_kernel_ms'].items() if v>0.08})
except Exception as e:
    print('$name', 'ERR', e, open('$O/bench_$name.err').read()[-300:])
PY
done

#!/bin/bash
# compact level pass in both arithmetic back ends: parity, the north-star configuration (10 x 1 M forward,
# sub-node prediction off) and the single-frame alt_flags leg
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_cx}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_arith.py tests/test_gpu_raht.py tests/test_gpu_tile.py tests/test_gpu_batches.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
for f in 1 0; do
for dir in forward inverse; do
  GPCC_F64=$f timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --subnode 0 --frames 10 --direction $dir > $O/bench_f64_${f}_$dir.json 2> $O/bench_f64_${f}_$dir.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_f64_${f}_$dir.json').read().strip().splitlines()[-1])
    r=d['roofline']
    k='forward_kernel_ms' if '$dir'=='forward' else 'inverse_kernel_ms'
    print('f64=$f $dir ms_per_step', d['ms_per_step'], {a:round(v,3) for a,v in r[k].items() if v>0.02})
except Exception as e:
    print('f64=$f $dir ERR', e, open('$O/bench_f64_${f}_$dir.err').read()[-300:])
PY
done
done

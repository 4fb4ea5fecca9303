#!/bin/bash
# the RDOQ stage with descriptor granules: parity of the sub-node encoder across qp / batches, headline + qp sweep
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_rdoq}; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_raht.py tests/test_gpu_batches.py tests/test_gpu_arith.py tests/test_gpu_pipe.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
for q in 34 28 22 16 10; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --qp $q > $O/bench_qp$q.json 2> $O/bench_qp$q.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_qp$q.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('qp$q ms_per_step', d['ms_per_step'], 'fwd', {k:round(v,3) for k,v in r['forward_kernel_ms'].items() if v>0.2}, 'inv', {k:round(v,3) for k,v in r['inverse_kernel_ms'].items() if v>0.2}, d['config']['roundtrip_decoder_equals_encoder_recon'])
except Exception as e:
    print('qp$q ERR', e, open('$O/bench_qp$q.err').read()[-300:])
PY
done

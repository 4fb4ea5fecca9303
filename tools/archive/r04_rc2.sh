#!/bin/bash
# recolour with the subtree kernel and the two trees built together; the bench's new legs
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_rc2}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_recolour.py tests/test_abi.py -m gpu -x -q ) > $O/pytest_rc.log 2>&1; tail -n 5 $O/pytest_rc.log
timeout 600 python tools/recolour_time.py > $O/recolour_time.txt 2>&1; tail -n 4 $O/recolour_time.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
print('qp_sweep', json.dumps(d.get('qp_sweep')))
print('recolour', json.dumps(d.get('recolour'))[:600])
print('pred passes', d['predicting'].get('encoder_pass_statistics'))
PY

#!/bin/bash
# recolour with the reference's k-d trees on the MI355X: parity tests, timing of the 1 M clouds; the headline with
# the level-by-level decoder beside the one walked across levels
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_rc}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_recolour.py -m gpu -x -q ) > $O/pytest_rc.log 2>&1; tail -n 5 $O/pytest_rc.log
timeout 600 python tools/recolour_time.py > $O/recolour_time.txt 2>&1; tail -n 4 $O/recolour_time.txt
for cfg in "pipe:GPCC_PIPE=1" "lvl:GPCC_PIPE=0"; do
  name=${cfg%%:*}; envs=$(echo ${cfg#*:} | tr ',' ' ')
  env $envs timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_$name.json 2> $O/bench_$name.err
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

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r03_sleep; mkdir -p $OUT
for v in main sleep8 sleep16 sleep32 sleep64; do
  if [ $v = main ]; then unset GPCC_LIB_PATH; else export GPCC_LIB_PATH=$PWD/mpeg-pcc-tmc13_amd/exp_$v.so; fi
  for f in 1 10; do
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile --frames $f --direction forward > $OUT/${v}_f$f.json 2>/dev/null
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile --frames $f --direction both > $OUT/${v}_b$f.json 2>/dev/null
  done
  python3 - <<PY
import json
r=[]
for f in (1,10):
    a=json.loads(open('$OUT/${v}_f%d.json'%f).read().strip().splitlines()[-1])['ms_per_step']
    b=json.loads(open('$OUT/${v}_b%d.json'%f).read().strip().splitlines()[-1])['ms_per_step']
    r.append('%d slices: fwd %.2f inv %.2f'%(f,a,b-a))
print('$v', ' | '.join(r))
PY
done

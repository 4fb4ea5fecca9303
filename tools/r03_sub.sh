#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_sub}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_pipe.py tests/test_gpu_full_size.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for f in 2 5 10 20; do
  for chunks in 1 0; do
    for dir in forward both; do
      GPCC_SUB_CHUNKS=$chunks timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile --frames $f --direction $dir > $OUT/c${chunks}_${dir}_$f.json 2> $OUT/c${chunks}_${dir}_$f.err
    done
  done
done
python3 - <<PY
import json
for f in (2,5,10,20):
    row=[]
    for chunks in (1,0):
        for d in ('forward','both'):
            try:
                j=json.loads(open('$OUT/c%d_%s_%d.json'%(chunks,d,f)).read().strip().splitlines()[-1])
                row.append('chunks=%d %s %.2f ms'%(chunks,d,j['ms_per_step']))
            except Exception as e:
                row.append('chunks=%d %s ERR'%(chunks,d))
    print(f, ' | '.join(row))
PY

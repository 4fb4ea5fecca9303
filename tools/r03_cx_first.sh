#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r03_cx1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_tile.py tests/test_gpu_slice_driver.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for sub in 0; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --subnode $sub --frames 10 --direction forward > $OUT/fwd10_sub$sub.json 2> $OUT/fwd10_sub$sub.err
  GPCC_CX=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --subnode $sub --frames 10 --direction forward > $OUT/fwd10_sub${sub}_old.json 2> $OUT/fwd10_sub${sub}_old.err
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --subnode 0 --frames 1 > $OUT/both1_sub0.json 2> $OUT/both1_sub0.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --subnode 0 --frames 10 --direction both > $OUT/both10_sub0.json 2> $OUT/both10_sub0.err
tail -c 1500 $OUT/fwd10_sub0.json; tail -c 300 $OUT/fwd10_sub0.err

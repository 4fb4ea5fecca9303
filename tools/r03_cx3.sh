#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_cx3}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_tile.py tests/test_gpu_slice_driver.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --subnode 0 --frames 10 --direction forward"
GPCC_PROFILE_LEVELS=1 timeout 300 $B > $OUT/fwd10_levels.json 2> $OUT/fwd10_levels.err
timeout 300 $B > $OUT/fwd10.json 2> $OUT/fwd10.err
CMD="$B --no-profile"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/pmc_sq -name '*.db') > $OUT/pmc_summary.txt 2>&1
find $OUT -name '*.db' -size +20M -delete
python3 - <<PY
import json
for f in ('fwd10_levels.json','fwd10.json'):
    d=json.loads(open('$OUT/'+f).read().strip().splitlines()[-1])
    ks=d['roofline'].get('forward_kernel_ms',{})
    print(f, d['ms_per_step'], ' '.join('%s=%.3f'%(k.replace('cx_level_enc','L'),ks[k]) for k in sorted(ks)))
PY
head -14 $OUT/pmc_summary.txt

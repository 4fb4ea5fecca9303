#!/bin/bash
# PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE: separate runs) of the default-flag batch: inverse of 10 slices, forward of 32
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_pmc_batch}
mkdir -p $OUT
for cfg in "10 inverse" "32 forward"; do
  set -- $cfg
  tag=f$1_$2
  B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile --frames $1 --direction $2"
  ( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/sq_$tag -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/sq_$tag.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/fe_$tag -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/fe_$tag.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/wr_$tag -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/wr_$tag.log 2>&1 )
  python tools/pmc_summary.py $(find $OUT/sq_$tag $OUT/fe_$tag $OUT/wr_$tag -name '*.db') > $OUT/pmc_$tag.txt 2>&1
  find $OUT -name '*.db' -delete
  head -16 $OUT/pmc_$tag.txt
done

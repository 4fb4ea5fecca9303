#!/bin/bash
# rocprofv3 kernel stats + PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE: separate runs) of the legs beside
# RAHT: LoD build + lifting, the predicting transform, recolouring.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_extras}
mkdir -p $OUT
for leg in lift pred recolour; do
  case $leg in
    lift) B="python tools/lift_time.py";;
    pred) B="python tools/pred_time.py";;
    recolour) B="python tools/recolour_time.py";;
  esac
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt_$leg -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/kt_$leg.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq_$leg -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_sq_$leg.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch_$leg -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_fetch_$leg.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_write_$leg -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_write_$leg.log 2>&1 )
  python tools/pmc_summary.py $(find $OUT/pmc_sq_$leg $OUT/pmc_fetch_$leg $OUT/pmc_write_$leg -name '*.db') > $OUT/pmc_summary_$leg.txt 2>&1
  find $OUT -name '*.db' -delete
  find $OUT/kt_$leg -name '*kernel_trace.csv' -delete
  tail -3 $OUT/kt_$leg.log
done
ls -la $OUT | head -30

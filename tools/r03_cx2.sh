#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r03_cx2
mkdir -p $OUT
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --subnode 0 --frames 10 --direction forward"
GPCC_PROFILE_LEVELS=1 timeout 300 $B > $OUT/fwd10_levels.json 2> $OUT/fwd10_levels.err
GPCC_LIB_PATH=$PWD/mpeg-pcc-tmc13_amd/exp_nosearch.so GPCC_PROFILE_LEVELS=1 timeout 300 $B > $OUT/fwd10_nosearch.json 2> $OUT/fwd10_nosearch.err
CMD="$B --no-profile"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write -name '*.db') > $OUT/pmc_summary.txt 2>&1
find $OUT -name '*.db' -size +20M -delete
find $OUT/kt -name '*kernel_stats.csv' | head -1 | xargs cat | head -20

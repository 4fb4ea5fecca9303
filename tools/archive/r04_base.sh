#!/bin/bash
# Round 4 baseline at HEAD: the GPU tier, the default bench line, per-level times of the
# headline, kernel trace + PMC of the headline configuration (default flags, one 1 M lidar frame).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_base}
mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
  echo "pytest rc $?" >> $OUT/pytest.log
  tail -4 $OUT/pytest.log
fi
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.err
GPCC_PROFILE_LEVELS=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 > $OUT/bench_levels.json 2> $OUT/bench_levels.err
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-profile"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write -name '*.db') > $OUT/pmc_summary.txt 2>&1
find $OUT -name '*.db' -delete
find $OUT/kt -name '*kernel_stats.csv' | head -1 | xargs -r head -25

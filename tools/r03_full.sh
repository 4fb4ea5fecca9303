#!/bin/bash
# Full GPU tier + the default bench line + profiles of the north-star configuration.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_full}
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-profile --subnode 0 --frames 10 --direction forward"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write -name '*.db') > $OUT/pmc_summary.txt 2>&1
find $OUT -name '*.db' -delete
tail -c 400 $OUT/bench_default.err

#!/bin/bash
# Round 4 at HEAD: the whole GPU tier, the default bench line, rocprofv3 kernel trace + PMC passes (separate) of the
# headline configuration, kernel traces of the north-star configuration and of recolour.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_final}
mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
  echo "pytest rc $?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 200 $OUT/bench_default.err
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-profile"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write -name '*.db') > $OUT/pmc_summary.txt 2>&1
B2="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-profile --subnode 0 --frames 10 --direction forward"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt10 -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $B2" > $GRAFT_REPO_ROOT/$OUT/kt10.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/ktrc -o kt -- bash -c "cd $GRAFT_REPO_ROOT && python tools/recolour_time.py" > $GRAFT_REPO_ROOT/$OUT/ktrc.log 2>&1 )
find $OUT -name '*.db' -delete
for d in kt kt10 ktrc; do f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -12 "$f" | cut -c1-160; done
# randomised differential stress against the compiled reference, bounded
( timeout 200 python tests/stress/stress_cx_batch.py 7710000 90 ) > $OUT/stress_cx_batch.txt 2>&1; tail -n 1 $OUT/stress_cx_batch.txt
( ALLFLAGS=1 timeout 200 python tests/stress/stress_cx_batch.py 7720000 90 ) > $OUT/stress_cx_batch_allflags.txt 2>&1; tail -n 1 $OUT/stress_cx_batch_allflags.txt
( timeout 200 python tests/stress/stress_raht.py 7730000 ) > $OUT/stress_raht.txt 2>&1; tail -n 1 $OUT/stress_raht.txt

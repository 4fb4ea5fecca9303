#!/bin/bash
# Round 4 at HEAD: the whole GPU tier, the default bench line, rocprofv3 kernel traces (headline; inter-frame RAHT),
# bounded randomised stress.  (The PMC passes of the headline are those of tools/r04_final.sh at 0bfc555: the headline's
# kernels have not changed since.)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_final2}
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 200 $OUT/bench_default.err
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-profile"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o kt -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1 )
( cd /tmp && GPCC_INTER_N=1000000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/ktinter -o kt -- bash -c "cd $GRAFT_REPO_ROOT && python tools/raht_inter_time.py" > $GRAFT_REPO_ROOT/$OUT/ktinter.log 2>&1 )
for d in kt ktinter; do f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -14 "$f" | cut -c1-170; done
( timeout 100 python tests/stress/stress_raht_inter_gpu.py 7740000 70 ) > $OUT/stress_raht_inter_gpu.txt 2>&1; tail -n 2 $OUT/stress_raht_inter_gpu.txt
( timeout 80 python tests/stress/stress_cx_batch.py 7710000 50 ) > $OUT/stress_cx_batch.txt 2>&1; tail -n 1 $OUT/stress_cx_batch.txt

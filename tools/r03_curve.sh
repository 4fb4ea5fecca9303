#!/bin/bash
# Batch-throughput curve of the default-flag (sub-node prediction) RAHT path and PMC passes
# for the 10 x 1M forward workload.  Run on the GPU box via gpurun.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r03_curve
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for dir in forward both; do
  for f in 1 2 5 10 20 32; do
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --frames $f --direction $dir \
      > $OUT/curve_${dir}_${f}.json 2> $OUT/curve_${dir}_${f}.err
  done
done
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile --frames 10 --direction forward"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write -name '*.db') > $OUT/pmc_summary.txt 2>&1
find $OUT -name '*.db' -size +20M -delete
tail -c 600 $OUT/curve_forward_10.json

#!/bin/bash
# rocprofv3 passes of the bench legs at HEAD -- kernel trace (--stats) and the PMC counters in passes of
# their own (SQ_*, FETCH_SIZE, WRITE_SIZE: gpurun refuses --pmc together with the trace domains), then
# tools/pmc_json.py turns the .db files into profiles-ready summaries.
#   bash tools/pmc.sh [out-dir-name] [workloads...]      workloads: headline fwd10_sub0 fwd10_sub1 legs
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc}
shift || true
WL=${@:-headline fwd10_sub0 fwd10_sub1 legs}
mkdir -p $OUT
COMMON="--steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-profile"
for w in $WL; do
  case $w in
    headline)   B="python bench.py $COMMON" ;;
    fwd10_sub0) B="python bench.py $COMMON --frames 10 --direction forward --subnode 0" ;;
    fwd10_sub1) B="python bench.py $COMMON --frames 10 --direction forward --subnode 1" ;;
    legs)       B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile --legs lifting,predicting,recolour,raht_inter" ;;
  esac
  R=$GRAFT_REPO_ROOT
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$w -o kt -- bash -c "cd $R && $B" > $R/$OUT/kt_$w.log 2>&1 )
  if [ "${KT_ONLY:-0}" = 1 ]; then
    find $OUT/kt_$w -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_$w.csv \; 2>/dev/null
    rm -rf $OUT/kt_$w
    continue
  fi
  ( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $R/$OUT/sq_$w -o sq -- bash -c "cd $R && $B" > $R/$OUT/sq_$w.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/fe_$w -o f -- bash -c "cd $R && $B" > $R/$OUT/fe_$w.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/wr_$w -o w -- bash -c "cd $R && $B" > $R/$OUT/wr_$w.log 2>&1 )
  python tools/pmc_summary.py $(find $OUT/sq_$w $OUT/fe_$w $OUT/wr_$w -name '*.db') > $OUT/pmc_$w.txt 2>&1
  python tools/pmc_json.py $w $(find $OUT/fe_$w $OUT/wr_$w -name '*.db') > $OUT/traffic_$w.json 2> $OUT/traffic_$w.err
  # the kernel-trace summary (csv) of the same command
  find $OUT/kt_$w -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_$w.csv \; 2>/dev/null
  find $OUT -name '*.db' -delete
  rm -rf $OUT/kt_$w $OUT/sq_$w $OUT/fe_$w $OUT/wr_$w
  head -12 $OUT/pmc_$w.txt
done

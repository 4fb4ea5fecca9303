# Round 6: texture-addresser / L1 counters of the north-star forward (10 x 1 M, sub-node off), default build and variants (separate --pmc passes)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=gpurun_out/r06_ta
mkdir -p $OUT
for v in ${VARIANTS:-base ary3}; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  i=0
  for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"; do
    i=$((i+1))
    ( cd /tmp && env $L timeout 300 rocprofv3 --pmc $set -d $R/$OUT/p_${v}_$i -o p -- bash -c "cd $R && python tools/fwd10_time.py 10 3 0" > $R/$OUT/p_${v}_$i.log 2>&1 )
  done
  python tools/pmc_summary.py $(find $OUT -path "*p_${v}_*" -name '*.db') > $OUT/ta_$v.txt 2>&1
  grep -A24 "cx_level_kernel" $OUT/ta_$v.txt | head -30
  find $OUT -name '*.db' -delete
done

# Round 6: pivots per step of the neighbour search of cx_level_kernel (GPCC_CX_SEARCH_ARY=3: a third of the window per step), exp/ build:
# north-star forward (10 x 1 M, sub-node off) per build, CRCs of the outputs
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_ary
for v in ${VARIANTS:-base ary3 base ary3}; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  echo "$v $(env $L timeout 300 python tools/fwd10_time.py 10 10 0 2>/dev/null | tail -1)" | tee -a gpurun_out/r06_ary/ab.txt
done

# Round 6: points per wavefront-tile of the tree build (GPCC_TILE_POINTS) and the grid cap of the compact pass's count / emit launches
# (GPCC_CX_TGRID_CAP), exp/ builds: the north-star forward per build
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_tile
for v in base tile512 tile256 tgrid tile512g base tile512; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  echo "$v $(env $L python tools/fwd10_time.py 10 10 0,1 2>/dev/null | tail -1)" | tee -a gpurun_out/r06_tile/ab.txt
done

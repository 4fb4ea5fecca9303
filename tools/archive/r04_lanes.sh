#!/bin/bash
# LoD device tier: slices in flight (GPCC_LOD_LANES) x workgroups per lane (GPCC_LOD_GRID), 5 x 1 M slices
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r04_lanes}; mkdir -p $O
for cfg in "4 256" "5 256" "5 192" "5 128" "5 160"; do
  set -- $cfg
  GPCC_LOD_LANES=$1 GPCC_LOD_GRID=$2 timeout 300 python tools/lift_time.py > $O/lanes$1_grid$2.txt 2>&1
  echo "lanes $1 grid $2: $(grep -o '"encode_ms": [0-9.]*, "decode_ms": [0-9.]*, "lod_build_alone_ms": [0-9.]*, "lod_build_ms_per_Mpoint": [0-9.]*, "value": [0-9.]*' $O/lanes$1_grid$2.txt)"
done

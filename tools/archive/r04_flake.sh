#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_flake}; mkdir -p $O
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_tile.py -m gpu -q -x > $O/tile_$i.log 2>&1; echo "tile alone $i rc $? $(tail -n 1 $O/tile_$i.log | cut -c1-80)"
done
AMD_LOG_LEVEL=1 timeout 600 python -m pytest tests/test_gpu_raht_inter.py tests/test_gpu_recolour.py tests/test_gpu_regions.py tests/test_gpu_tile.py -m gpu -q -x > $O/seq.log 2>&1; echo "sequence rc $? $(tail -n 1 $O/seq.log | cut -c1-80)"
grep -n -i "fault\|abort\|error" $O/seq.log | head -5

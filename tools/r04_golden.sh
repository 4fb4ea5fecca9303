#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r04_golden}; mkdir -p $O
timeout 120 python -m pytest tests/test_golden_raht_inter.py tests/test_golden_recolour.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?"; tail -n 2 $O/pytest.log

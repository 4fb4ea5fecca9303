#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r04_probe}; mkdir -p $O
timeout 40 python tools/pin_hazard_probe.py pinned 400 > $O/pinned.log 2>&1; echo "pinned rc $? $(tail -n 1 $O/pinned.log | cut -c1-120)"
timeout 40 python tools/pin_hazard_probe.py pageable 400 > $O/pageable.log 2>&1; echo "pageable rc $? $(tail -n 1 $O/pageable.log | cut -c1-120)"

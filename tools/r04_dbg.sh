#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_dbg}; mkdir -p $O
cat > /tmp/repro.py <<'PY'
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as g; g.load_package()
import numpy as np
from mpeg_pcc_tmc13_amd import context, lift_params, lod_params, synth
from mpeg_pcc_tmc13_amd.params import set_qp_regions
ctx = context(0)
xyz, attrs = synth.dense_cloud(20000, seed=61, bits=9)
lo, hi = xyz.min(axis=0), xyz.max(axis=0)
regs = [(tuple(lo), tuple((lo+hi)//2), (-5, 2))]
lp = lod_params()
print("plain", flush=True)
ctx.lift_encode_attr(lp, lift_params([len(xyz)], qp=34), xyz, attrs)
print("regions", flush=True)
lf2 = set_qp_regions(lift_params([len(xyz)], qp=34), regs)
co, rec, lcp, idx = ctx.lift_encode_attr(lp, lf2, xyz, attrs)
print("ok", co[:3], flush=True)
PY
AMD_LOG_LEVEL=1 HIP_LAUNCH_BLOCKING=1 timeout 300 python /tmp/repro.py > $O/repro.log 2>&1; echo "rc $?" >> $O/repro.log; tail -n 30 $O/repro.log
( time timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_regions.py -k "not region" ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log

#!/bin/bash
# Round 6: parity subset + bench + round profile of the sub-node path with the coarse-level sweep
set -u
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-r06_sweep}
bash tools/r06_gpu.sh $N "tests/test_gpu_raht.py tests/test_gpu_batches.py tests/test_gpu_regions.py tests/test_gpu_tile.py" "on|GPCC_SWEEP=1|--no-extras --steps 10 --no-cpu-baseline" "on22|GPCC_SWEEP=1|--no-extras --steps 10 --no-cpu-baseline --qp 22" > gpurun_out/${N}_stdout.txt 2>&1
GPCC_LIB_PATH=exp/libgpcc_subprof.so python tools/sweep_prof.py 1 forward > gpurun_out/$N/prof_fwd.txt 2>&1
GPCC_LIB_PATH=exp/libgpcc_subprof.so python tools/sweep_prof.py 1 inverse > gpurun_out/$N/prof_inv.txt 2>&1
grep -E "passed|failed|FAILED" gpurun_out/$N/pytest.log
python - <<PY
import json
for n in ("on","on22"):
    d=json.loads(open("gpurun_out/$N/bench_%s.json"%n).read().splitlines()[0])
    print(n, d["value"], d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], d["roofline"]["forward_kernel_ms"], d["roofline"]["inverse_kernel_ms"])
PY
grep -E "level +([6789]|1[0-9])|per round" gpurun_out/$N/prof_fwd.txt gpurun_out/$N/prof_inv.txt

#!/bin/bash
# Third matrix: the call returns after 0.26 ms in a slow step too (tools/r05_stall_probe4.py: in_call_ms), so the
# 10-80 ms are on the device's side of the queue, between dispatches.  schedule_kernel (284 B/lane), the level
# kernels of three components (12-36 B/lane) and the sub-node kernels (28-164 B/lane) use scratch: ROCr's scratch
# policy (reclaim / use-once above a limit) under a few settings.  Three runs per setting.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_stall_env3
run() {
  tag=$1; shift
  for k in a b c; do
    env "$@" PROBE_REPS=8 timeout 120 python tools/r05_stall_probe4.py > gpurun_out/r05_stall_env3/$tag.$k.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_stall_env3/$tag.$k.json"))["slow"]
    print("$tag.$k", "slow steps:", len(d), [(x["subnode"], x["slices"], x["direction"], x["wall_ms"], x["in_call_ms"]) for x in d][:8])
except Exception as e:
    print("$tag.$k", "ERR", e)
PY
  done
}
run no_reclaim HSA_NO_SCRATCH_RECLAIM=1
run no_async_reclaim HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
run single_limit_4g HSA_SCRATCH_SINGLE_LIMIT=4000000000 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4000000000
run default X=1

#!/bin/bash
# Second matrix for the one-in-ten slow step (HIP_FORCE_DEV_KERNARG did not hold up on repetition: 4 and 3 slow steps
# with it against 1 and 2 without, profiles/r05_stall_env_matrix.txt): the host-side every-N-commands mechanisms of the
# HIP runtime and the way it waits.  Two runs per setting; tools/r05_stall_probe3.py prints the steps above 3x the median.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_stall_env2
run() {
  tag=$1; shift
  for k in a b; do
    env HIP_FORCE_DEV_KERNARG=0 "$@" timeout 200 python tools/r05_stall_probe3.py > gpurun_out/r05_stall_env2/$tag.$k.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_stall_env2/$tag.$k.json"))
    print("$tag.$k", "slow steps:", len(d), [(x["subnode"], x["slices"], x["direction"], x["wall_ms"]) for x in d][:8])
except Exception as e:
    print("$tag.$k", "ERR", e)
PY
  done
}
run batch_100k DEBUG_CLR_MAX_BATCH_SIZE=100000
run active_wait ROC_ACTIVE_WAIT_TIMEOUT=200000
run signal_pool ROC_SIGNAL_POOL_SIZE=4096
run no_direct AMD_DIRECT_DISPATCH=0
run cpu_wait ROC_CPU_WAIT_FOR_SIGNAL=0
run default X=1

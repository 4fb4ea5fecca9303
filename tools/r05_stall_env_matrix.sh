#!/bin/bash
# Which runtime setting does the one-in-ten slow step of the compact-pass batches depend on?  tools/r05_stall_probe3.py
# (6 repetitions of the bench's sequence of batches; prints the slow steps) under a few settings of the HIP / ROCr runtime.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_stall_env
run() {
  tag=$1; shift
  env "$@" timeout 200 python tools/r05_stall_probe3.py > gpurun_out/r05_stall_env/$tag.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_stall_env/$tag.json"))
    print("$tag", "slow steps:", len(d), [(x["subnode"], x["slices"], x["direction"], x["wall_ms"]) for x in d][:8])
except Exception as e:
    print("$tag", "ERR", e)
PY
}
run default X=1
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run host_kernarg HIP_FORCE_DEV_KERNARG=0
run no_interrupt HSA_ENABLE_INTERRUPT=0
run no_sdma HSA_ENABLE_SDMA=0
run one_queue GPU_MAX_HW_QUEUES=1
run default2 X=1

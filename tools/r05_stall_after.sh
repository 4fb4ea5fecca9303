#!/bin/bash
# After the library / bench prefer kernel arguments in device memory: the stall probe at the new default and with
# the old setting forced, twice each (tools/r05_stall_probe3.py prints the steps above 3x the median).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_stall_after
for tag in new_default_a host_kernarg_a new_default_b host_kernarg_b; do
  case $tag in host*) E="HIP_FORCE_DEV_KERNARG=0";; *) E="X=1";; esac
  env $E timeout 200 python tools/r05_stall_probe3.py > gpurun_out/r05_stall_after/$tag.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_stall_after/$tag.json"))
    print("$tag", "slow steps:", len(d), [(x["subnode"], x["slices"], x["direction"], x["wall_ms"]) for x in d][:8])
except Exception as e:
    print("$tag", "ERR", e)
PY
done

#!/bin/bash
# After schedule_kernel lost its scratch: tools/r05_stall_probe4.py five times at the defaults, then twice with the
# setting that made the slow steps frequent before (HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0: 8, 8, 10 of them).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_stall_after2
for tag in default.a default.b default.c default.d default.e no_async_reclaim.a no_async_reclaim.b; do
  case $tag in no_async*) E="HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0";; *) E="X=1";; esac
  env $E PROBE_REPS=8 timeout 120 python tools/r05_stall_probe4.py > gpurun_out/r05_stall_after2/$tag.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_stall_after2/$tag.json"))["slow"]
    print("$tag", "slow steps:", len(d), [(x["subnode"], x["slices"], x["direction"], x["wall_ms"], x["in_call_ms"]) for x in d][:8])
except Exception as e:
    print("$tag", "ERR", e)
PY
done

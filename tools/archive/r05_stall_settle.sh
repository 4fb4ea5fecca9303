#!/bin/bash
# The slow steps are the first one or two timed steps after a batch was created (67 + 38 of 133 recorded ones are steps
# 0 and 1), and 6479 consecutive steps of the worst leg show none (tools/r05_stall_probe5.py): does a pause between the
# two warm-up calls and the timed steps remove them?  Three runs with a 0.3 s pause, three without.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_stall_settle
for tag in settle.a plain.a settle.b plain.b settle.c plain.c; do
  case $tag in settle*) E="PROBE_SETTLE=0.3";; *) E="PROBE_SETTLE=0";; esac
  env $E PROBE_REPS=8 timeout 200 python tools/r05_stall_probe4.py > gpurun_out/r05_stall_settle/$tag.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_stall_settle/$tag.json"))["slow"]
    print("$tag", "slow steps:", len(d), [(x["subnode"], x["slices"], x["direction"], x["step"], x["wall_ms"]) for x in d][:8])
except Exception as e:
    print("$tag", "ERR", e)
PY
done

#!/bin/bash
# Round 6: the sweep's level threshold (GPCC_SWEEP_PARENTS) against the per-level kernels, headline frame and a batch
set -u
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-r06_sweep_thr}
mkdir -p gpurun_out/$N
for thr in 0 32 256 1024 8192; do
  for fr in 1 10; do
    if [ $thr = 0 ]; then E="GPCC_SWEEP=0"; else E="GPCC_SWEEP=1 GPCC_SWEEP_PARENTS=$thr"; fi
    env $E python bench.py --no-extras --steps 10 --no-cpu-baseline --frames $fr > gpurun_out/$N/b_${thr}_$fr.json 2> gpurun_out/$N/b_${thr}_$fr.err
  done
done
python - <<PY
import json
for thr in (0, 32, 256, 1024, 8192):
    for fr in (1, 10):
        d=json.loads(open("gpurun_out/$N/b_%d_%d.json"%(thr,fr)).read().splitlines()[0])
        r=d["roofline"]
        print(thr, fr, d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], "fwd launches", r["forward_launches"], "fwd kernels %.3f" % sum(r["forward_kernel_ms"].values()), "inv kernels %.3f" % sum(r["inverse_kernel_ms"].values()), {k:v for k,v in r["forward_kernel_ms"].items() if "sweep" in k}, {k:v for k,v in r["inverse_kernel_ms"].items() if "sweep" in k})
PY

cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_pipe
for v in 0 1; do
  env GPCC_PIPE=$v python bench.py --no-extras --steps 10 --no-cpu-baseline > gpurun_out/r06_pipe/b_$v.json 2>/dev/null
done
python - <<PY
import json
for v in (0,1):
    d=json.loads(open("gpurun_out/r06_pipe/b_%d.json"%v).read().splitlines()[0])
    r=d["roofline"]
    print("pipe",v, d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], "inv %.3f" % sum(r["inverse_kernel_ms"].values()), r["inverse_kernel_ms"], r["inverse_launches"])
PY

cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_sleep
for v in base 0 1 16; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_sleep$v.so"; fi
  for q in 34 22; do
    env $L python bench.py --no-extras --steps 10 --no-cpu-baseline --qp $q > gpurun_out/r06_sleep/b_${v}_$q.json 2>/dev/null
  done
done
python - <<PY
import json
for v in ("base","0","1","16"):
    for q in (34,22):
        d=json.loads(open("gpurun_out/r06_sleep/b_%s_%d.json"%(v,q)).read().splitlines()[0])
        r=d["roofline"]
        print("sleep",v,"qp",q, d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], "fwd %.3f inv %.3f" % (sum(r["forward_kernel_ms"].values()), sum(r["inverse_kernel_ms"].values())))
PY

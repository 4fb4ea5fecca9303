# Round 6: round-robin polling of the awaited granules in raht_level_sub_kernel (GPCC_SUB_POLL_RR, exp/libgpcc_rr.so) against
# lowest-first: headline frame at qp 34 / 22 / textured, dense C=3 frame, a batch of ten
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_rr
for v in base rr; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  env $L python bench.py --no-extras --steps 10 --no-cpu-baseline > gpurun_out/r06_rr/${v}_lidar34.json 2>/dev/null
  env $L python bench.py --no-extras --steps 5 --no-cpu-baseline --qp 22 > gpurun_out/r06_rr/${v}_lidar22.json 2>/dev/null
  env $L python bench.py --no-extras --steps 5 --no-cpu-baseline --cloud dense > gpurun_out/r06_rr/${v}_dense.json 2>/dev/null
  env $L python bench.py --no-extras --steps 5 --no-cpu-baseline --frames 10 > gpurun_out/r06_rr/${v}_batch10.json 2>/dev/null
done
python - <<PY
import json
for w in ("lidar34","lidar22","dense","batch10"):
    for v in ("base","rr"):
        d=json.loads(open("gpurun_out/r06_rr/%s_%s.json"%(v,w)).read().splitlines()[0]); r=d["roofline"]
        print(w, v, d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], "fwd %.3f inv %.3f" % (sum(r["forward_kernel_ms"].values()), sum(r["inverse_kernel_ms"].values())))
PY

# Round 6: length of the s_sleep in the sub-node kernels' polling loop (compile-time GPCC_SUB_SLEEP; exp/ builds),
# one frame and a batch of ten
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_sleep
for v in base 0 1 16 64; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_sleep$v.so"; fi
  [ $v = base ] || [ -f exp/libgpcc_sleep$v.so ] || continue
  for fr in 1 10; do
    env $L python bench.py --no-extras --steps 5 --no-cpu-baseline --frames $fr > gpurun_out/r06_sleep/f_${v}_$fr.json 2>/dev/null
  done
done
python - <<PY
import json, os
for v in ("base","0","1","16","64"):
    for fr in (1,10):
        p="gpurun_out/r06_sleep/f_%s_%d.json"%(v,fr)
        if not os.path.exists(p): continue
        d=json.loads(open(p).read().splitlines()[0])
        r=d["roofline"]
        print("sleep",v,"frames",fr, d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], "fwd %.3f inv %.3f" % (sum(r["forward_kernel_ms"].values()), sum(r["inverse_kernel_ms"].values())))
PY

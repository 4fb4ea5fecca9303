# Round 6: the waiting iteration of lod_subsample_distance_kernel without the decision block when no neighbour has arrived
# (GPCC_LOD_IDLE_FAST; exp/libgpcc_lodidle0.so = without): per-level times, one lane, and the lifting leg (four lanes)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_lodidle
for v in ${VARIANTS:-lodidle0 base lodidle0 base}; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  env $L python tools/lod_level_times.py > gpurun_out/r06_lodidle/lv_$v.json 2>/dev/null
  env $L python tools/lift_time.py > gpurun_out/r06_lodidle/lift_$v.json 2>/dev/null
  python - <<PY | tee -a gpurun_out/r06_lodidle/ab.txt
import json
v="$v"
d=json.loads(open("gpurun_out/r06_lodidle/lv_%s.json"%v).read().splitlines()[-1])["kernel_ms_per_build"]
sub=[d.get("lod_subsample@%02d"%i,0) for i in range(10)]
l=json.loads(open("gpurun_out/r06_lodidle/lift_%s.json"%v).read().splitlines()[-1])
print(v, "subsample per level", [round(x,2) for x in sub], "sum %.2f" % sum(sub), "| lifting leg: lod ms/Mpt", l["lod_build_ms_per_Mpoint"], "enc", l["encode_ms"], "ok", l["roundtrip_decoder_equals_encoder_recon"])
PY
done
GPCC_LIB_PATH=${TESTLIB:-} timeout 900 python -m pytest tests/test_gpu_lod.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee -a gpurun_out/r06_lodidle/ab.txt

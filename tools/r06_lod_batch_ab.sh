# Round 6: granules polled per round trip in lod_subsample_distance_kernel (GPCC_LOD_BATCH, exp/ builds): per-level times, one lane,
# and the lifting leg (four lanes)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_lodb
for v in base lodb19w2 lodb19w3 lodb7w3; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  env $L python tools/lod_level_times.py > gpurun_out/r06_lodb/lv_$v.json 2>/dev/null
  env $L python tools/lift_time.py > gpurun_out/r06_lodb/lift_$v.json 2>/dev/null
done
python - <<PY
import json
for v in ("base","lodb19w2","lodb19w3","lodb7w3"):
    d=json.loads(open("gpurun_out/r06_lodb/lv_%s.json"%v).read().splitlines()[-1])["kernel_ms_per_build"]
    sub=[d.get("lod_subsample@%02d"%i,0) for i in range(10)]
    l=json.loads(open("gpurun_out/r06_lodb/lift_%s.json"%v).read().splitlines()[-1])
    print(v, "subsample per level", [round(x,2) for x in sub], "sum %.2f" % sum(sub), "| lifting leg: lod ms/Mpt", l["lod_build_ms_per_Mpoint"], "enc", l["encode_ms"], "ok", l["roundtrip_decoder_equals_encoder_recon"])
PY

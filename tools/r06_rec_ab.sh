# Round 6: block records in the per-level sub-node kernels (GPCC_REC=0 / 1; default: batches only), one frame and batches
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_rec
timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_batches.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06_rec/pytest_rec1.log 2>&1
GPCC_REC=1 timeout 900 python -m pytest tests/test_gpu_raht.py tests/test_gpu_batches.py tests/test_gpu_tile.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06_rec/pytest_rec_forced.log 2>&1
tail -2 gpurun_out/r06_rec/pytest_rec1.log gpurun_out/r06_rec/pytest_rec_forced.log
for v in 0 1; do
  for fr in 1 4 10; do
    env GPCC_REC=$v python bench.py --no-extras --steps 5 --no-cpu-baseline --frames $fr > gpurun_out/r06_rec/b_${v}_$fr.json 2>/dev/null
  done
done
python - <<PY
import json
for v in (0,1):
    for fr in (1,4,10):
        d=json.loads(open("gpurun_out/r06_rec/b_%d_%d.json"%(v,fr)).read().splitlines()[0])
        r=d["roofline"]
        print("rec",v,"frames",fr, d["ms_per_step"], d["config"]["roundtrip_decoder_equals_encoder_recon"], "fwd %.3f inv %.3f" % (sum(r["forward_kernel_ms"].values()), sum(r["inverse_kernel_ms"].values())), {k:round(x,3) for k,x in r["forward_kernel_ms"].items() if x>0.2}, {k:round(x,3) for k,x in r["inverse_kernel_ms"].items() if x>0.2})
PY

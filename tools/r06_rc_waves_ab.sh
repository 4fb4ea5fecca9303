# Round 6: wavefronts per SIMD the recolour search kernels are compiled for (GPCC_RC_WAVES, exp/ builds): the recolour leg per build
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_rcw
for v in rcw0 rcw6 rcw8 rcw0 rcw6 rcw8; do
  GPCC_LIB_PATH=exp/libgpcc_$v.so python bench.py --legs recolour --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); r=d.get('recolour', d)
print('$v', json.dumps({k:r[k] for k in r if k in ('call_ms','kernel_ms')}))" | tee -a gpurun_out/r06_rcw/ab.txt
done
GPCC_LIB_PATH=exp/libgpcc_rcw8.so timeout 600 python -m pytest tests/test_gpu_recolour.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/r06_rcw/ab.txt

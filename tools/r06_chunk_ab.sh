# Round 6: rounds per class and turn of the dependency kernels' claims (GPCC_SUB_CHUNK: 1 = rounds interleave over the eight classes /
# XCDs) and a plain store in front of the write-through store of a granule (GPCC_SUB_DSTORE), exp/ builds: the hand-off
# micro-benchmark, then the headline per build
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_chunk
timeout 120 exp/handoff 2>&1 | grep -E "sc1 / load sc1|plain / load sc1" | tee gpurun_out/r06_chunk/handoff.txt
for v in base ck4 ck16 ck64 ck1d ck16d ck64d ck256d base; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  env $L timeout 300 python bench.py --no-extras --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], 'Mpts/s', d['ms_per_step'], 'ms; fwd', r['forward_kernel_ms'].get('level_sub_lossy'), 'inv', r['inverse_kernel_ms'].get('level_sub_synth'), 'roundtrip', d['config']['roundtrip_decoder_equals_encoder_recon'])" | tee -a gpurun_out/r06_chunk/ab.txt
done

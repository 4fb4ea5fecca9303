# Round 6: granules polled per lane and iteration in raht_level_sub_kernel (GPCC_SUB_POLL_N, exp/ builds; default 1 = the lowest awaited one):
# the headline per build, then the 10 x 1 M forward with sub-node prediction
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_polln
for v in base pn2 pn3 pn4 base pn2; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  env $L timeout 300 python bench.py --no-extras --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], 'Mpts/s', d['ms_per_step'], 'ms; fwd', r['forward_kernel_ms'].get('level_sub_lossy'), 'inv', r['inverse_kernel_ms'].get('level_sub_synth'), 'roundtrip', d['config']['roundtrip_decoder_equals_encoder_recon'])" | tee -a gpurun_out/r06_polln/ab.txt
done
for v in base pn2 pn4; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  echo "$v $(env $L timeout 300 python tools/fwd10_time.py 10 10 1 2>/dev/null | tail -1)" | tee -a gpurun_out/r06_polln/ab.txt
done

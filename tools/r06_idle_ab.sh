# Round 6: the waiting iteration of raht_level_sub_kernel without the mailbox loop (unless the mailbox gained a child) and without the
# 12-way look-up of the polled granule's row (unless the slot changed) -- GPCC_SUB_IDLE_FAST, exp/ builds: headline, then 10 x 1 M forward
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_idle
for v in ${VARIANTS:-idle0 idle1 idle0 idle1}; do
  L="GPCC_LIB_PATH=exp/libgpcc_$v.so"
  env $L timeout 300 python bench.py --no-extras --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], 'Mpts/s', d['ms_per_step'], 'ms; fwd', r['forward_kernel_ms'].get('level_sub_lossy'), 'inv', r['inverse_kernel_ms'].get('level_sub_synth'), 'roundtrip', d['config']['roundtrip_decoder_equals_encoder_recon'])" | tee -a gpurun_out/r06_idle/ab.txt
done
for v in ${VARIANTS10:-idle0 idle1}; do
  L="GPCC_LIB_PATH=exp/libgpcc_$v.so"
  echo "$v $(env $L timeout 300 python tools/fwd10_time.py 10 10 1 2>/dev/null | tail -1)" | tee -a gpurun_out/r06_idle/ab.txt
done

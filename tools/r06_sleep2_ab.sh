# Round 6: s_sleep of the waiting iteration again, now that the iteration is short (GPCC_SUB_SLEEP 4 = default, 1, 0): headline, qp 22 and the
# textured field (total encoder / decoder kernel time per transform from tools/raht_level_times.py)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_sleep2
for v in ${VARIANTS:-base s1 s0 base s1 s0}; do
  if [ $v = base ]; then L=""; else L="GPCC_LIB_PATH=exp/libgpcc_$v.so"; fi
  for args in "34" "22" "34 24"; do
    env $L python tools/raht_level_times.py $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
enc=sum(v for n,v in k.items() if 'lossy' in n); dec=sum(v for n,v in k.items() if 'synth' in n)
print('$v', 'qp/noise $args', 'encoder %.3f decoder %.3f ms' % (enc, dec), 'roundtrip', d['roundtrip'])" | tee -a gpurun_out/r06_sleep2/ab.txt
  done
done

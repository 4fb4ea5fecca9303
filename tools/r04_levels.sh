#!/bin/bash
# per-level kernel times of the sub-node kernels (one 1 M lidar frame)
O=gpurun_out/${1:-r04_levels}; mkdir -p $O
for cfg in ${CFGS:-"f64_lvl:GPCC_F64=1,GPCC_PIPE=0"}; do
  name=${cfg%%:*}; envs=$(echo ${cfg#*:} | tr ',' ' ')
  env $envs GPCC_PROFILE_LEVELS=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
r=d['roofline']
f=r['forward_kernel_ms']; i=r['inverse_kernel_ms']
print('$name', d['ms_per_step'])
for li in range(17,-1,-1):
    a=f.get('level_sub_lossy@%02d'%li); b=i.get('level_sub_synth@%02d'%li)
    if a or b: print('  li %2d  enc %.3f  dec %.3f'%(li, a or 0, b or 0))
PY
done

import sys
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g; g.load_package()
import numpy as np, torch
from mpeg_pcc_tmc13_amd import synth, raht_params, context
kind=sys.argv[1]
xyz,attrs=(synth.lidar_cloud(1000000,seed=1) if kind=='lidar' else synth.dense_cloud(1000000,seed=1,bits=10))
morton,attrs,_=synth.sort_by_morton(xyz,attrs)
p=raht_params(qp=34,subnode=True,search_range=2500 if kind=='lidar' else 50000)
ctx=context(0)
for i in range(2):
    co,rec=ctx.raht_forward(p,morton,attrs)

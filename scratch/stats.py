import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g; g.load_package()
import numpy as np, torch
from mpeg_pcc_tmc13_amd import synth, raht_params, context, _lib
kind=sys.argv[1]
xyz,attrs=(synth.lidar_cloud(1000000,seed=1) if kind=='lidar' else synth.dense_cloud(1000000,seed=1,bits=10))
morton,attrs,_=synth.sort_by_morton(xyz,attrs)
p=raht_params(qp=34,subnode=True,search_range=2500 if kind=='lidar' else 50000)
ctx=context(0)
lib=_lib.load()
out=(C.c_ulonglong*16)()
co,rec=ctx.raht_forward(p,morton,attrs)
lib.gpcc_debug_stats(out,1)
co,rec=ctx.raht_forward(p,morton,attrs)
lib.gpcc_debug_stats(out,1)
names=['waves(rounds)','iters total','idle iters','lookback episodes','lookback steps','max iters/wave','max lb steps/group','groups with lb','max lb distance']
for n,v in zip(names,out): print(kind,n,v)

"""Randomised differential stress of the predicting transform on the device against the
oracle (python tests/stress/stress_pred.py <seed> [seconds]): random clouds, LoD parameters,
tools, QP layers, region offsets, quantisation-weight shares incl. wrapping ones.  Found the
unsigned-overload rounding of computeQuantizationWeights (tests/test_oracle_pred.py:lidar_qnw_wrap)."""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import __graft_entry__ as g; g.load_package()
import numpy as np
import lod_helpers as lh
from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth, context
ctx=context(0)
t0=time.time(); cases=0
base=int(sys.argv[1]); budget=float(sys.argv[2]) if len(sys.argv)>2 else 100
for seed in range(100000):
    rng=np.random.default_rng(base+seed)
    big = seed % 5 == 0
    n=int(rng.integers(20000,120000)) if big else int(rng.integers(1,4000))
    kind=int(rng.integers(3))
    if kind==0: xyz,attrs=synth.random_cloud(n,seed=int(rng.integers(1<<30)),bits=int(rng.integers(2,11)),dup_fraction=float(rng.choice([0.0,0.2])))
    elif kind==1: xyz,attrs=synth.dense_cloud(n,seed=int(rng.integers(1<<30)),bits=int(rng.integers(6,11)))
    else: xyz,attrs=synth.lidar_cloud(n,seed=int(rng.integers(1<<30)))
    c=attrs.shape[1]
    bitdepth=int(rng.choice([8,10])) if c==3 else int(rng.choice([8,16]))
    if bitdepth>8: attrs=(attrs.astype(np.int64)*int(rng.integers(1,(1<<bitdepth)//256+1))).astype(np.int32)
    levels=int(rng.integers(1,14))
    lp=lod_params(levels=levels, decimation=int(rng.integers(3)), dist2=int(rng.integers(0,3)), neighbours=int(rng.integers(1,4)), lifting=False,
                  distribution=bool(rng.integers(2)), bias=tuple(int(x) for x in rng.integers(1,4,size=3)), inter_range=int(rng.choice([4,64,128,1100000])),
                  intra_range=int(rng.choice([0,8,64,1100000])), sampling_period=int(rng.integers(1,6)), blend=bool(rng.integers(2)))
    lp.intra_lod_prediction_skip_layers=int(rng.choice([0,2,0x7fffffff]))
    lod=ctx.lod_build(lp,xyz)
    direct=int(rng.integers(0,4)); dis=bool(rng.integers(2)) and direct>0
    qnw=tuple(int(v) for v in rng.choice([[0,0,0],[16,8,4],[25,12,12],[130,90,40],[5,0,7]]))
    nl=int(rng.integers(1,4)); layers=[(int(rng.integers(4,52+6*(bitdepth-8))), int(rng.integers(-3,4))) for _ in range(nl)]
    kw=dict(bitdepth=bitdepth, layers=layers, max_levels=levels, quant_neigh_weight=qnw, icp=bool(rng.integers(2)), threshold=int(rng.choice([0,4,64])), avg_disabled=dis)
    qp_off=None
    if rng.integers(3)==0:
        qp_off=np.zeros((len(xyz),2),np.int32); m=rng.random(len(xyz))<0.3; qp_off[m]=(int(rng.integers(-20,20)),int(rng.integers(-5,5)))
    pp=pred_params(lod["npl"], direct=direct, **kw)
    v,rec,icp,modes=lh.oracle_pred(True,pp,lod,attrs=attrs,qp_off=qp_off)
    got=ctx.pred_inverse(pp,lod["nc"],lod["ni"],lod["w"],lod["indexes"],v,icp=icp,qp_off=qp_off)
    assert np.array_equal(got,rec), ("dec",seed,n,kw,direct)
    pp0=pred_params(lod["npl"], direct=0, **dict(kw,avg_disabled=False))
    v0,rec0,icp0,_=lh.oracle_pred(True,pp0,lod,attrs=attrs,qp_off=qp_off)
    gv,grec,gicp=ctx.pred_forward(pp0,lod["nc"],lod["ni"],lod["w"],lod["indexes"],attrs,qp_off=qp_off)
    assert np.array_equal(gv,v0) and np.array_equal(grec,rec0), ("enc",seed,n,kw)
    if c==3 and kw["icp"]: assert np.array_equal(gicp,icp0), ("icp",seed)
    cases+=1
    if time.time()-t0>budget: break
print("pred stress ok", cases, "cases", round(time.time()-t0,1),"s")

import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as g; g.load_package()
from mpeg_pcc_tmc13_amd import synth, recolour_params, context
import oracle_loader as ol
ctx = context(0); ctx.set_profiling(True)
for kind, n, scale in (('dense', 1_000_000, 0.5), ('lidar', 1_000_000, 0.25)):
    xyz, a = (synth.dense_cloud if kind=='dense' else synth.lidar_cloud)(n, seed=1)
    tgt = np.unique(np.rint(xyz.astype(np.float64)*scale).astype(np.int32), axis=0)
    p = recolour_params()
    ctx.recolour(p, xyz, a, tgt, scale=scale); ctx.kernel_times()
    t0=time.time(); got = ctx.recolour(p, xyz, a, tgt, scale=scale); t1=time.time()
    kt = ctx.kernel_times()
    t2=time.time(); ref = ol.ref().recolour(p, xyz, a, tgt, scale=scale); t3=time.time()
    print(kind, 'ns', len(xyz), 'nt', len(tgt), 'device call %.1f ms'%((t1-t0)*1e3), {k: round(v[0],3) for k,v in kt.items()}, 'ref %.2f s'%(t3-t2), 'agree %.4f'%np.all(got==ref,axis=1).mean())

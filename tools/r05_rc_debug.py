import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as g; g.load_package()
from mpeg_pcc_tmc13_amd import synth, recolour_params, context
import oracle_loader as ol
ctx = context(0)
for kind, n, scale, kw in (('dense', 20000, 0.37, {}), ('dense', 1000000, 0.5, {}), ('dense', 6000, 0.5, {}), ('lidar', 6000, 0.25, {}), ('dense', 20000, 0.37, dict(k_bwd=3))):
    xyz, a = (synth.dense_cloud(n, seed=3, bits=9) if kind=='dense' else synth.lidar_cloud(n, seed=3))
    tgt = np.unique(np.rint(xyz.astype(np.float64)*scale).astype(np.int32), axis=0)
    p = recolour_params(bitdepth=8, **kw)
    try:
        got = ctx.recolour(p, xyz, a, tgt, scale=scale)
        ora = ol.oracle().recolour(p, xyz, a, tgt, scale=scale)
        print(kind, n, scale, kw, 'equal', np.array_equal(got, ora), 'differing rows', int((got!=ora).any(axis=1).sum()), flush=True)
    except Exception as e:
        print(kind, n, scale, kw, 'EXC', repr(e)[:200], flush=True)
        break

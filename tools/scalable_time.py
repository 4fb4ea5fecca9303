#!/usr/bin/env python3
"""LoD build + lifting of one slice with and without aps.scalable_lifting_enabled_flag (GPU box):
host-tier one-call entry, per-kernel times from the context's profiler."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
g.load_package()
from mpeg_pcc_tmc13_amd import context, lift_params, lod_params, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = context(0)
out = {}
for kind in ("dense", "lidar"):
    xyz, attrs = synth.dense_cloud(n, seed=7, bits=10) if kind == "dense" else synth.lidar_cloud(n, seed=7)
    for scalable in (0, 1):
        lp = lod_params()
        lp.scalable_lifting_enabled_flag = scalable
        lp.max_neigh_range_minus1 = 5
        ms = []
        for rep in range(4):
            lf = lift_params([len(xyz)], qp=34, lcp=(attrs.shape[1] == 3))
            t = time.perf_counter()
            ctx.lift_encode_attr(lp, lf, xyz, attrs)
            ms.append((time.perf_counter() - t) * 1e3)
        ctx.set_profiling(True)
        lf = lift_params([len(xyz)], qp=34, lcp=(attrs.shape[1] == 3))
        ctx.lift_encode_attr(lp, lf, xyz, attrs)
        kt = ctx.kernel_times()
        ctx.set_profiling(False)
        agg = {}
        for name, (t_ms, launches) in kt.items():
            key = name.rstrip("0123456789").rstrip("_")
            agg[key] = round(agg.get(key, 0.0) + t_ms, 3)
        out[f"{kind}_scalable{scalable}"] = dict(points=len(xyz), lods=int(lf.num_lods), host_call_ms=round(min(ms[1:]), 2),
                                                  kernels_ms=dict(sorted(agg.items(), key=lambda kv: -kv[1])[:8]))
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Phase cycles of the compact level pass (experiment build -DGPCC_CX_PROF):
    GPCC_LIB_PATH=<exp .so> python tools/cx_prof.py [frames]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

g.load_package()
import torch
from mpeg_pcc_tmc13_amd import _lib, context, raht_params, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sub = 0
dev = torch.device("cuda:0")
ctx = context(0)
p = raht_params(qp=34, subnode=bool(sub), search_range=2500)
fr = []
for f in range(frames):
    xyz, a = synth.lidar_cloud(1_000_000, seed=1 + f)
    m, a, _ = synth.sort_by_morton(xyz, a)
    fr.append((m, a))
offs = np.concatenate([[0], np.cumsum([len(m) for m, a in fr])]).astype(np.int64)
d_m = torch.from_numpy(np.concatenate([m for m, a in fr])).to(dev)
src = torch.from_numpy(np.concatenate([a for m, a in fr]).reshape(-1)).to(dev)
d_a = torch.empty_like(src)
d_c = torch.zeros(int(offs[-1]), dtype=torch.int32, device=dev)
lib = _lib.load()
out = (C.c_ulonglong * 20)()
ctx.set_morton_bits(54)
for it in range(3):
    d_a.copy_(src)
    ctx.dev_raht_forward(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 1)
    ctx.synchronize()
    if it == 0:
        lib.gpcc_debug_cx_prof(out, 1)
lib.gpcc_debug_cx_prof(out, 0)
v = [int(x) for x in out]
tiles = max(1, v[12])
names = ["setup", "search", "pred gather", "fwd butterflies", "quant+rdoq stats", "hypotheses", "look-back wait",
         "decisions+coeff stores", "inverse+stores"]
tot = sum(v[:9])
print(f"tiles {tiles // 2} per forward; memtime ticks per tile (100 MHz clock):")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {v[i] / tiles:9.1f}  {100.0 * v[i] / tot:5.1f} %")
print(f"  look-back spins per waiting tile {v[10] / max(1, v[11]):.2f}; tiles that looked back {v[13] / tiles:.3f}")

n2, n3, dist, nospin = v[16], v[17], v[18], v[19]
print(f"  resolved by a closed word: {n2} (spins {v[14] / max(1, n2):.1f} each), by a final word: {n3} (spins {v[15] / max(1, n3):.1f} each); "
      f"mean distance back {dist / max(1, n2 + n3):.2f} tiles; resolved at the first look {nospin}")

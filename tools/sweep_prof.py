#!/usr/bin/env python3
"""Round / iteration statistics of the sub-node kernels (experiment build -DGPCC_SUB_PROF):
    GPCC_LIB_PATH=<exp .so> python tools/sub_prof.py [frames] [forward|inverse]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g

g.load_package()
import torch
from mpeg_pcc_tmc13_amd import _lib, context, raht_params, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
direction = sys.argv[2] if len(sys.argv) > 2 else "forward"
dev = torch.device("cuda:0")
ctx = context(0)
p = raht_params(qp=int(os.environ.get("QP", "34")), subnode=True, search_range=50000)
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
out = (C.c_ulonglong * (16 + 32 * 20))()
ctx.set_morton_bits(54)
import time
for it in range(3):
    d_a.copy_(src)
    ctx.dev_raht_forward(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 1)
    ctx.synchronize()
    if direction == "inverse":
        lib.gpcc_debug_sub_prof(out, 1)
        t0 = time.perf_counter()
        ctx.dev_raht_inverse(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 1)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    elif it == 1:
        lib.gpcc_debug_sub_prof(out, 1)
        t0 = time.perf_counter()
lib.gpcc_debug_sub_prof(out, 0)
v = [int(x) for x in out]
rounds = max(1, v[0])
print(f"{direction}, {frames} slices: rounds {rounds}; per round: prologue {v[1] / rounds:.0f} ticks, loop {v[2] / rounds:.0f} ticks, "
      f"stage ticks {[round(v[3 + i] / rounds) for i in range(4)]}, iterations {v[7] / rounds:.2f}, idle iterations {v[8] / rounds:.2f}")
for li in range(0, 21):
    r = v[16 + li * 4]
    if r:
        print(f"  level {li:2d}: rounds {r:8d} prologue {v[17 + li * 4] / r:9.0f} loop {v[18 + li * 4] / r:9.0f} iterations {v[19 + li * 4] / r:6.2f}"
              f" idle {v[148 + li * 6] / r:6.2f} stage ticks X {v[144 + li * 6] / r:8.0f} P {v[145 + li * 6] / r:7.0f} Z {v[146 + li * 6] / r:7.0f} W {v[147 + li * 6] / r:7.0f}")
        b = 336 + li * 10
        if v[b] or v[b + 1]:
            pe, we = max(1, v[b]), max(1, v[b + 1])
            print(f"            (P) x {v[b] / r:5.2f} per round: to butterflies' end {v[b + 2] / pe:6.0f}, quantise + descriptor {v[b + 3] / pe:6.0f} cycles each | "
                  f"(W) x {v[b + 1] / r:5.2f}: to inverse butterflies' end {v[b + 4] / we:6.0f}, mailbox + granule store {v[b + 5] / we:6.0f}, rec stores {v[b + 6] / we:6.0f}")

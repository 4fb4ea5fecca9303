#!/usr/bin/env python3
"""RAHT forward of F x 1 M-point S-lidar slices in one batch (the north-star configuration), per sub-node flag: median
wall time, kernel times (HIP events) and a checksum of the outputs -- for A/B runs of experiment builds:
    GPCC_LIB_PATH=<exp .so> python tools/fwd10_time.py [frames] [steps] [subs, e.g. 0 or 0,1]"""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

g.load_package()
import torch
from mpeg_pcc_tmc13_amd import context, raht_params, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
subs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
dev = torch.device("cuda:0")
# one stream carries torch's copies and the library's launches (as bench.py does)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = context(0, stream=stream.cuda_stream)
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
ctx.set_morton_bits(54)
ctx.reserve(int(offs[-1]), frames, 1)
out = {}
for sub in subs:
    p = raht_params(qp=34, subnode=bool(sub), search_range=2500)

    def fwd():
        d_a.copy_(src)
        ctx.dev_raht_forward(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 1)

    for _ in range(2):
        fwd()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fwd()
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    ctx.set_profiling(True)
    ctx.kernel_times()
    for _ in range(3):
        fwd()
    torch.cuda.synchronize(dev)
    kt = ctx.kernel_times()
    ctx.set_profiling(False)
    out[f"subnode_{sub}"] = {
        "ms_median": round(ts[len(ts) // 2] * 1e3, 3), "ms_min": round(ts[0] * 1e3, 3), "ms_max": round(ts[-1] * 1e3, 3),
        "kernel_ms": {k: round(v[0] / 3, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
        "crc_coeffs": zlib.crc32(d_c.cpu().numpy().tobytes()), "crc_recon": zlib.crc32(d_a.cpu().numpy().tobytes())}
print(json.dumps(out))

#!/usr/bin/env python3
"""Per tree level: kernel time of the RAHT forward and inverse transform of the headline frame (1 M-point S-lidar, reference default
flags) -- per-level timer names (GPCC_PROFILE_LEVELS=1), HIP events.   [GPCC_LIB_PATH=..] python tools/raht_level_times.py [qp] [noise]"""
import json
import os
import sys

import numpy as np

os.environ.setdefault("GPCC_PROFILE_LEVELS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
import torch  # noqa: E402
from mpeg_pcc_tmc13_amd import context, raht_params, synth  # noqa: E402

qp = int(sys.argv[1]) if len(sys.argv) > 1 else 34
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = context(0, stream=stream.cuda_stream)
xyz, a = synth.lidar_cloud(1_000_000, seed=1) if len(sys.argv) < 3 else synth.lidar_cloud(1_000_000, seed=1, refl_noise=int(sys.argv[2]))
m, a, _ = synth.sort_by_morton(xyz, a)
n = len(m)
offs = np.array([0, n], dtype=np.int64)
d_m = torch.from_numpy(m).to(dev)
src = torch.from_numpy(a.reshape(-1).astype(np.int32)).to(dev)
d_a = torch.empty_like(src)
d_d = torch.empty_like(src)
d_c = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.set_morton_bits(54)
p = raht_params(qp=qp, search_range=2500)


def step():
    d_a.copy_(src)
    ctx.dev_raht_forward(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 1)
    ctx.dev_raht_inverse(p, offs, d_m.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), 1)


for _ in range(3):
    step()
torch.cuda.synchronize(dev)
ctx.set_profiling(True)
ctx.kernel_times()
reps = 5
for _ in range(reps):
    step()
torch.cuda.synchronize(dev)
kt = ctx.kernel_times()
ctx.set_profiling(False)
out = {k: round(v[0] / reps, 4) for k, v in sorted(kt.items())}
print(json.dumps({"roundtrip": bool(torch.equal(d_a, d_d)), "kernel_ms": out}))

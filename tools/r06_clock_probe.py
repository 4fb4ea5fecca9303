#!/usr/bin/env python3
"""Round 6 probe: does the headline's latency-bound walk run at a reduced shader clock because the chip is mostly asleep?
Times the forward / inverse transform of the headline frame alone and with a stream of large matrix products running beside
it on a second stream (something that keeps the chip's clocks up)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
import torch  # noqa: E402
from mpeg_pcc_tmc13_amd import context, raht_params, synth  # noqa: E402

dev = torch.device("cuda:0")
main = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
ctx = context(0, stream=main.cuda_stream)
xyz, a = synth.lidar_cloud(1_000_000, seed=1)
m, a, _ = synth.sort_by_morton(xyz, a)
n = len(m)
offs = np.array([0, n], dtype=np.int64)
d_m = torch.from_numpy(m).to(dev)
src = torch.from_numpy(a.reshape(-1).copy()).to(dev)
d_a = torch.empty_like(src)
d_c = torch.zeros(n, dtype=torch.int32, device=dev)
d_r = torch.empty_like(src)
p = raht_params(qp=34, subnode=True, search_range=2500)
ctx.set_morton_bits(54)


def step():
    d_a.copy_(src)
    ctx.dev_raht_forward(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 1)
    ctx.dev_raht_inverse(p, offs, d_m.data_ptr(), d_r.data_ptr(), d_c.data_ptr(), 1)


def timed(k=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    main.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


res = {"alone_ms": round(timed(), 3)}
for size, label in ((1024, "side_matmul_1k"), (4096, "side_matmul_4k")):
    A = torch.randn(size, size, device=dev, dtype=torch.float16)
    B = torch.randn(size, size, device=dev, dtype=torch.float16)
    stop = False
    with torch.cuda.stream(side):
        for _ in range(2000 if size == 1024 else 120):  # a few hundred ms of work queued on the side stream
            C = A @ B
    res[label + "_ms"] = round(timed(), 3)
    side.synchronize()
res["alone_again_ms"] = round(timed(), 3)
print(json.dumps(res))

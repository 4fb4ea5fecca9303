"""Probe of the hazard behind the one abort of the GPU tier (profiles/r04_gpu_tier_and_stress_final.txt): uploads out of
PAGEABLE numpy memory (the runtime pins the pages on the fly, read-only, and keeps the pins), the arrays freed, then
downloads into fresh pageable memory that may land on the same addresses.  argv[1] = "pageable" | "pinned" (the form
tests/conftest.py and bench.py use now), argv[2] = iterations.  Exit code 0 = survived."""
import sys
import numpy as np
import torch

mode, iters = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
for it in range(iters):
    a = rng.integers(0, 100, size=int(rng.integers(1, 60)) * 100_000, dtype=np.int32)
    t = torch.from_numpy(a)
    if mode == "pinned":
        t = t.pin_memory()
    d = t.to(dev)
    del a, t
    out = torch.full((int(rng.integers(1, 60)) * 100_000,), it, dtype=torch.int32, device=dev)
    h = out.cpu()
    assert int(h[0]) == it and int(h[-1]) == it
    del d, out, h
torch.cuda.synchronize()
print(mode, "survived", iters, "iterations")

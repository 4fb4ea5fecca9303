#!/usr/bin/env python3
"""VERDICT r04 weak #10 / round 5's batch curve: ONE step in ten of the sub-node-off forward takes tens of milliseconds
longer than its neighbours (36.5 against 1.9 / 3.4 ms at 5 / 10 slices; the driver's round-4 run saw 22 ms in the
inverse).  This probe runs a few hundred steps and, for every step, records the host's wall time, the GPU's own time
between two events around the step, and the library's allocation events (arena growth, pool misses, pinned staging) --
so a slow step is attributed to the library, to the GPU or to the host side of the process.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import gc
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    from mpeg_pcc_tmc13_amd import _lib, context, raht_params
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(0, stream=stream.cuda_stream)
    frames = [bench.make_frame("lidar", 1_000_000, seed=1 + i) for i in range(10)]
    out = {}
    for nf, sub, direction in ((10, 0, "forward"), (5, 0, "forward"), (10, 0, "inverse"), (10, 1, "forward")):
        p = raht_params(qp=34, subnode=bool(sub), search_range=2500)
        b = bench.Batch(torch, dev, ctx, frames[:nf], p)
        b.forward()
        fn = b.forward if direction == "forward" else b.inverse
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        steps = 150 if sub == 0 else 40
        rows = []
        ev = (C.c_longlong * 4)()
        gc_before = gc.get_count()
        for i in range(steps):
            lib.gpcc_debug_alloc_events(ctx._h, ev)
            e0 = list(ev)
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record(stream)
            fn()
            t1 = time.perf_counter()          # the host has enqueued everything
            z.record(stream)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            lib.gpcc_debug_alloc_events(ctx._h, ev)
            rows.append((t2 - t0, a.elapsed_time(z) / 1e3, t1 - t0, [int(ev[k] - e0[k]) for k in range(4)]))
        wall = sorted(r[0] for r in rows)
        med = wall[len(wall) // 2]
        slow = [(i, r) for i, r in enumerate(rows) if r[0] > 3 * med + 0.002]
        out[f"{direction}_{nf}x1M_sub{sub}"] = {
            "steps": steps, "median_ms": round(med * 1e3, 3), "max_ms": round(wall[-1] * 1e3, 3),
            "slow_steps": [{"step": i, "wall_ms": round(r[0] * 1e3, 3), "gpu_ms": round(r[1] * 1e3, 3),
                            "host_enqueue_ms": round(r[2] * 1e3, 3), "alloc_events": r[3]} for i, r in slow][:12],
            "num_slow": len(slow), "gc_counts_before": gc_before, "gc_counts_after": gc.get_count()}
        del b
    print(json.dumps(out))


if __name__ == "__main__":
    main()

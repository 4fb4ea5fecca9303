#!/usr/bin/env python3
"""Third probe: WHICH kernel holds the 30-50 ms of the slow step (sub-node-off forward, 5 / 10 slices)?  The bench's
sequence (a default-flag curve first, then the sub-node-off one), every timed forward step with the library's per-kernel
HIP-event timers on; for a slow step the kernels' times of THAT step are printed."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    from mpeg_pcc_tmc13_amd import context, raht_params
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(0, stream=stream.cuda_stream)
    ctx.set_profiling(True)
    frames = [bench.make_frame("lidar", 1_000_000, seed=1 + i) for i in range(10)]
    out = []
    for rep in range(6):
        for sub in (1, 0):
            p = raht_params(qp=34, subnode=bool(sub), search_range=2500)
            for nf in (1, 2, 5, 10):
                b = bench.Batch(torch, dev, ctx, frames[:nf], p)
                for fn_name in ("forward", "inverse"):
                    fn = getattr(b, fn_name)
                    fn(); fn()
                    torch.cuda.synchronize(dev)
                    ctx.kernel_times()
                    ts = []
                    for i in range(10):
                        t0 = time.perf_counter()
                        fn()
                        torch.cuda.synchronize(dev)
                        dt = time.perf_counter() - t0
                        kt = ctx.kernel_times()
                        ts.append((dt, kt))
                    med = sorted(t for t, _ in ts)[5]
                    for i, (dt, kt) in enumerate(ts):
                        if dt > 3 * med + 0.003:
                            out.append({"rep": rep, "subnode": sub, "slices": nf, "direction": fn_name, "step": i,
                                        "wall_ms": round(dt * 1e3, 2), "median_ms": round(med * 1e3, 3),
                                        "kernels_ms": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:6]},
                                        "kernels_sum_ms": round(sum(v[0] for v in kt.values()), 3)})
                del b
    print(json.dumps(out))


if __name__ == "__main__":
    main()

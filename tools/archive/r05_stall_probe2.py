#!/usr/bin/env python3
"""Second probe of the one-in-ten 30 ms step of bench.py's batch curve (sub-node-off forward, 5 and 10 slices): the
bench's own forward_10m() run three times in a fresh process, nothing else around it; every curve point's median / max /
slowest step, and the library's allocation events around each call of the leg."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    from mpeg_pcc_tmc13_amd import _lib, context, raht_params
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(0, stream=stream.cuda_stream)

    def params_for(cloud, subnode, haar=False, qp=34):
        return raht_params(qp=qp, subnode=bool(subnode), search_range=2500 if cloud == "lidar" else 50000)
    frames = [bench.make_frame("lidar", 1_000_000, seed=1)]
    out = []
    ev = (C.c_longlong * 4)()
    for rep in range(3):
        lib.gpcc_debug_alloc_events(ctx._h, ev)
        e0 = list(ev)
        r = bench.forward_10m(torch, dev, ctx, params_for, frames)
        lib.gpcc_debug_alloc_events(ctx._h, ev)
        pts = []
        for k, v in r["batch_curve"].items():
            for p in v:
                worst = max(p["forward_ms_max"] / p["forward_ms"], p["inverse_ms_max"] / p["inverse_ms"])
                pts.append({"flags": k, "slices": p["slices"], "forward_ms": p["forward_ms"], "forward_ms_max": p["forward_ms_max"],
                            "forward_slowest_step": p["forward_slowest_step"], "inverse_ms": p["inverse_ms"],
                            "inverse_ms_max": p["inverse_ms_max"], "worst_ratio": round(worst, 2)})
        out.append({"rep": rep, "alloc_events": [int(ev[i] - e0[i]) for i in range(4)],
                    "slow_points": [p for p in pts if p["worst_ratio"] > 1.5], "all_points": pts})
    print(json.dumps(out))


if __name__ == "__main__":
    main()

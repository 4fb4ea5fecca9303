#!/usr/bin/env python3
"""A/B of the neighbour links (raht_links.hpp) on the MI355X: the headline step and the north-star configuration
(RAHT forward, 10 x 1M S-lidar slices) with both states of the sub-node flag; run once per state of GPCC_LINKS
(the library reads it once per process).  Prints one JSON object.

    GPCC_LINKS=0 python tools/r05_links_ab.py; GPCC_LINKS=1 python tools/r05_links_ab.py"""
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
    from mpeg_pcc_tmc13_amd import context, raht_params
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(0, stream=stream.cuda_stream)
    nslices = int(os.environ.get("AB_SLICES", "10"))
    frames = [bench.make_frame("lidar", 1_000_000, seed=1 + i) for i in range(nslices)]
    out = {"GPCC_LINKS": os.environ.get("GPCC_LINKS", "1")}
    for sub in (1, 0):
        p = raht_params(qp=34, subnode=bool(sub), search_range=2500)
        b1 = bench.Batch(torch, dev, ctx, frames[:1], p)

        def step():
            b1.forward()
            b1.inverse()
        med, mx = bench.timed_stats(torch, dev, step, 20, warmup=3)
        ok1 = b1.roundtrip_ok()
        ktf = bench.kernel_profile(torch, dev, ctx, b1.forward, 5)
        kti = bench.kernel_profile(torch, dev, ctx, b1.inverse, 5)
        out[f"headline_sub{sub}"] = {
            "ms_per_step_median": round(med * 1e3, 3), "ms_per_step_max": round(mx * 1e3, 3), "roundtrip": ok1,
            "Mpts": round(b1.n / med / 1e6, 2),
            "forward_kernel_ms": {k: round(v[0], 4) for k, v in sorted(ktf.items(), key=lambda kv: -kv[1][0])},
            "inverse_kernel_ms": {k: round(v[0], 4) for k, v in sorted(kti.items(), key=lambda kv: -kv[1][0])},
            "forward_launches": round(sum(v[1] for v in ktf.values()), 1)}
        del b1
        b = bench.Batch(torch, dev, ctx, frames, p)
        tf, tfm = bench.timed_stats(torch, dev, b.forward, 10, warmup=2)
        ti, tim = bench.timed_stats(torch, dev, b.inverse, 10, warmup=2)
        ok = b.roundtrip_ok()
        kt = bench.kernel_profile(torch, dev, ctx, b.forward, 3)
        out[f"forward_{nslices}x1M_sub{sub}"] = {
            "forward_ms_median": round(tf * 1e3, 3), "forward_ms_max": round(tfm * 1e3, 3),
            "inverse_ms_median": round(ti * 1e3, 3), "inverse_ms_max": round(tim * 1e3, 3), "roundtrip": ok,
            "forward_Mpts": round(b.n / tf / 1e6, 1),
            "kernel_ms": {k: round(v[0], 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
            "launches": round(sum(v[1] for v in kt.values()), 1)}
        del b
    ctx.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

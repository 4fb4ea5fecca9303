#!/usr/bin/env python3
"""A/B of the sub-node encoder's claim size (GPCC_SUB_CLAIM = rounds of 8 blocks a wavefront takes per claim at the
coarse levels, raht_subnode.hpp) on the MI355X: headline frame at qp 34 / 22 / 10, the textured frame, forward of
10 x 1M slices.  One process per setting (the library reads the switch once).  Prints one JSON object."""
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
    out = {"GPCC_SUB_CLAIM": os.environ.get("GPCC_SUB_CLAIM", "default"),
           "GPCC_SUB_CLAIM_PARENTS": os.environ.get("GPCC_SUB_CLAIM_PARENTS", "default")}
    frames = [bench.make_frame("lidar", 1_000_000, seed=1 + i) for i in range(10)]
    tex = [bench.make_frame("lidar", 1_000_000, seed=1, refl_noise=24)]
    for name, fr, qp in (("qp34", frames[:1], 34), ("qp22", frames[:1], 22), ("qp10", frames[:1], 10), ("textured_qp34", tex, 34)):
        p = raht_params(qp=qp, subnode=True, search_range=2500)
        b = bench.Batch(torch, dev, ctx, fr, p)
        tf, tfm = bench.timed_stats(torch, dev, b.forward, 10, warmup=2)
        ti, tim = bench.timed_stats(torch, dev, b.inverse, 10, warmup=2)
        kt = bench.kernel_profile(torch, dev, ctx, b.forward, 3)
        out[name] = {"forward_ms": round(tf * 1e3, 3), "forward_ms_max": round(tfm * 1e3, 3), "inverse_ms": round(ti * 1e3, 3),
                     "roundtrip": b.roundtrip_ok(), "Mpts_fwd_inv": round(b.n / (tf + ti) / 1e6, 2),
                     "level_sub_lossy_ms": round(kt.get("level_sub_lossy", (0, 0))[0], 3)}
        del b
    p = raht_params(qp=34, subnode=True, search_range=2500)
    b = bench.Batch(torch, dev, ctx, frames, p)
    tf, tfm = bench.timed_stats(torch, dev, b.forward, 6, warmup=2)
    b.inverse()
    out["forward_10x1M"] = {"forward_ms": round(tf * 1e3, 3), "forward_ms_max": round(tfm * 1e3, 3), "roundtrip": b.roundtrip_ok()}
    ctx.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

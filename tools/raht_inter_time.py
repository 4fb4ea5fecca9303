"""Times gpcc_raht_forward_inter / _inverse_inter on 1 M-point frames (host tier: host buffers in, host buffers out)
with the per-kernel times of the context's profiler; prints the CPU reference's time beside it when it is built."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from mpeg_pcc_tmc13_amd import RahtInterParams, context, raht_params, synth  # noqa: E402


def frame_of(xyz, attrs, rng, drop=0.1, jitter=4):
    keep = rng.random(len(xyz)) > drop
    x = np.clip(xyz + rng.integers(-1, 2, size=xyz.shape), 0, None)[keep].astype(np.int32)
    a = np.clip(attrs + rng.integers(-jitter, jitter + 1, size=attrs.shape), 0, 255)[keep].astype(np.int32)
    return synth.sort_by_morton(x, a)[:2]


def main():
    n = int(os.environ.get("GPCC_INTER_N", "1000000"))
    ctx = context(0)
    rng = np.random.default_rng(1)
    out = {}
    for kind in ("lidar", "dense"):
        if kind == "lidar":
            xyz, attrs = synth.lidar_cloud(n, seed=7)
            attrs = (attrs >> 8 if attrs.max() > 255 else attrs)[:, :1]
        else:
            xyz, attrs = synth.dense_cloud(n, seed=8, bits=10)
        morton, a, _ = synth.sort_by_morton(xyz, attrs)
        mref, aref = frame_of(xyz, attrs, rng)
        for sub, rdo, fest in ((0, 0, 0), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
            p = raht_params(subnode=bool(sub))
            ip = RahtInterParams(15, rdo, fest, 3)
            ctx.raht_forward_inter(p, ip, morton, a, mref, aref)
            # wall time without the profiler (with it the encoder's two candidates run one after the other)
            t0 = time.perf_counter()
            co, rec, modes, taps = ctx.raht_forward_inter(p, ip, morton, a, mref, aref)
            t1 = time.perf_counter()
            dec = ctx.raht_inverse_inter(p, ip, morton, co, a.shape[1], mref, aref, modes, taps)
            t2 = time.perf_counter()
            ctx.set_profiling(True)
            ctx.kernel_times()
            ctx.raht_forward_inter(p, ip, morton, a, mref, aref)
            kt_f = ctx.kernel_times()
            ctx.set_profiling(False)
            agg = {}
            for name, (ms, _) in kt_f.items():
                agg[name.split("@")[0]] = round(agg.get(name.split("@")[0], 0.0) + ms, 3)
            row = {"n": len(morton), "c": a.shape[1], "n_ref": len(mref), "forward_ms": round((t1 - t0) * 1e3, 2),
                   "inverse_ms": round((t2 - t1) * 1e3, 2), "decoder_equals_encoder": bool(np.array_equal(dec, rec)),
                   "modes": modes.tolist(), "taps": taps.tolist(), "nonzero": int((co != 0).sum()),
                   "forward_kernels_ms": dict(sorted(agg.items(), key=lambda kv: -kv[1])[:10])}
            if os.environ.get("GPCC_INTER_REF") == "1":
                import ctypes as C
                import oracle_loader as ol
                from test_oracle_raht_inter import run
                if ol.ref_available():
                    t0 = time.perf_counter()
                    rc, co_r, _, modes_r, taps_r = run(ol.ref().lib, "ref_raht_inter", p, True, morton, a, None, mref, aref, 15, rdo, fest, 3)
                    row["cpu_reference_forward_s"] = round(time.perf_counter() - t0, 3)
                    row["identical_to_reference"] = bool(np.array_equal(co, co_r) and np.array_equal(modes, modes_r) and np.array_equal(taps, taps_r))
            out[f"{kind}_sub{sub}_rdo{rdo}_fest{fest}"] = row
            print(kind, "subnode", sub, "decision", rdo, "estimated_taps", fest, json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

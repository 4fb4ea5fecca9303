#!/usr/bin/env python3
"""bench.py -- RAHT attribute-transform throughput on MI355X.

One STEP = one pass of the hot path over one batch of synthetic,
Morton-sorted frames resident in HBM: RAHT forward (encoder side:
coefficients + reconstruction) followed by RAHT inverse (decoder side) of
every frame of the batch.  At N=1 the workload is BASELINE.json configs[1]:
a 1M-point lidar-shaped cloud (S-lidar, 18-bit grid, reflectance, C=1),
flags of cfg/octree-raht-ctc-lossless-geom-lossy-attrs.yaml (qp 34, search
range 2500) and the reference's defaults for everything else, i.e.
raht_subnode_prediction_enabled_flag = 1 (TMC3.cpp:1307).  The same frames
with sub-node prediction switched off (blocks of a level independent) and
the lifting path (LoD build + lifting forward/inverse, configs[2] shape)
are reported in extra objects of the same line.  With N>1 every rank transforms its own frame(s) (weak scaling,
frames shard one-per-GPU) and the quantised coefficients are gathered on
rank 0 with one RCCL gather inside the timed region.

Prints ONE JSON line (see the driver contract).  value = points entering
the forward+inverse pass per second, whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per frame")
    ap.add_argument("--frames", type=int, default=1, help="frames (slices) per GPU per step")
    ap.add_argument("--cloud", choices=["lidar", "dense"], default="lidar")
    ap.add_argument("--qp", type=int, default=34)
    ap.add_argument("--subnode", type=int, default=1)
    ap.add_argument("--haar", type=int, default=0)
    ap.add_argument("--direction", choices=["both", "inverse"], default="both",
                    help="inverse: decoder only (coefficients prepared on the CPU outside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the alternative-flag and lifting legs")
    return ap.parse_args()


def make_frame(args, seed):
    from mpeg_pcc_tmc13_amd import synth
    if args.cloud == "lidar":
        xyz, attrs = synth.lidar_cloud(args.points, seed=seed)
        bits = 18
    else:
        bits = 10 if args.points <= 1_500_000 else 12
        xyz, attrs = synth.dense_cloud(args.points, seed=seed, bits=bits)
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    return morton, attrs, 3 * bits


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    from mpeg_pcc_tmc13_amd import context, raht_params

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL ("nccl") is the backend of every real run; GPCC_BENCH_BACKEND=gloo
    # exists only to exercise the multi-rank control flow on a one-GPU box
    # (several ranks share the device, the gather goes through host memory)
    backend = os.environ.get("GPCC_BENCH_BACKEND", "nccl")
    local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where collectives operate

    if args.haar:
        p = raht_params(qp=4, haar=True, chroma_offset=0, subnode=bool(args.subnode), search_range=2500)
    else:
        p = raht_params(qp=args.qp, subnode=bool(args.subnode),
                        search_range=2500 if args.cloud == "lidar" else 50000)

    # ---- synthetic frames of this rank, resident in HBM -------------------
    frames = [make_frame(args, seed=1 + rank * args.frames + f) for f in range(args.frames)]
    c = frames[0][1].shape[1]
    sizes = [len(f[0]) for f in frames]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offsets[-1])
    d_morton = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
    src = torch.from_numpy(np.concatenate([f[1] for f in frames]).reshape(-1)).to(dev)
    d_attrs = torch.empty_like(src)
    d_coeffs = torch.zeros(c * n, dtype=torch.int32, device=dev)
    d_dec = torch.empty_like(src)
    gathered = ([torch.empty_like(d_coeffs, device=xdev) for _ in range(world)]
                if (world > 1 and rank == 0) else None)

    stream = torch.cuda.current_stream(dev)
    ctx = context(local_rank, stream=stream.cuda_stream)
    ctx.set_morton_bits(frames[0][2])

    if args.direction == "inverse":
        # decoder-only run (e.g. CTC flags with sub-node prediction, whose lossy
        # forward is not on the device yet): coefficients from the CPU checker
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_loader as ol
        chk = ol.ref() if ol.ref_available() else ol.oracle()
        cos, recs = zip(*[chk.raht_forward(p, f[0], f[1]) for f in frames])
        d_coeffs.copy_(torch.from_numpy(np.concatenate(cos)).to(dev))
        d_attrs.copy_(torch.from_numpy(np.concatenate(recs).reshape(-1)).to(dev))

    def step():
        if args.direction == "both":
            d_attrs.copy_(src)  # the transform overwrites its input with the reconstruction
            ctx.dev_raht_forward(p, offsets, d_morton.data_ptr(), d_attrs.data_ptr(), d_coeffs.data_ptr(), c)
        ctx.dev_raht_inverse(p, offsets, d_morton.data_ptr(), d_dec.data_ptr(), d_coeffs.data_ptr(), c)
        if world > 1:
            dist.gather(d_coeffs if backend == "nccl" else d_coeffs.cpu(), gathered, dst=0)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # in-run sanity: decoder output == encoder reconstruction (the
    # reference's own conformance criterion)
    roundtrip_ok = bool(torch.equal(d_attrs, d_dec))

    ms_per_step = elapsed / args.steps * 1e3
    value = n * world * args.steps / elapsed / 1e6

    out = {
        "metric": "RAHT forward+inverse attribute-transform Mpoints/s (bit-exact vs CPU reference)",
        "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int64 (Q15 fixed point), int32 I/O",
        "data": "synthetic",
        "config": {
            "workload": f"RAHT {'forward+inverse' if args.direction == 'both' else 'inverse only'}, {args.frames}x{args.points}-point Morton-sorted "
                        f"S-{args.cloud} frame(s) per GPU, C={c}, "
                        + ("integer Haar qp 4" if args.haar else f"qp {args.qp}")
                        + f", raht_prediction=1, raht_subnode_prediction={int(args.subnode)}, raht_extension=1",
            "points_per_gpu_per_step": n, "frames_per_gpu": args.frames,
            "roundtrip_decoder_equals_encoder_recon": roundtrip_ok,
        },
    }

    if rank == 0 and world == 1:
        # ---- per-kernel durations: HIP events on the context's stream -----
        if not args.no_profile:
            ctx.set_profiling(True)
            ctx.kernel_times()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize(dev)
            kt = ctx.kernel_times()
            ctx.set_profiling(False)
            total_ms = sum(v[0] for v in kt.values())
            name, (dom_ms, dom_launches) = max(kt.items(), key=lambda kv: kv[1][0])
            # algorithmic bytes of one step: forward
            # (8 + 12 C) B/pt + inverse (8 + 8 C) B/pt; the level kernels
            # are where attributes and coefficients are consumed/produced,
            # so one step's launches of the dominant kernel are priced as
            # one pass over those bytes
            alg_bytes = n * (((8 + 12 * c) if args.direction == 'both' else 0) + (8 + 8 * c))
            dom_s = dom_ms / 1e3 / args.steps
            achieved = alg_bytes / dom_s / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": pmc_traffic(args, name),
                "traffic_collected_at_launch_us": pmc_traffic(args, name, "avg_launch_us_at_collection"),
                "launches_per_step": dom_launches / args.steps,
                "algorithmic_bytes_per_launch": round(alg_bytes * args.steps / dom_launches),
                "avg_launch_us": round(dom_ms * 1e3 / dom_launches, 2),
                "pipeline_achieved": round(alg_bytes / (ms_per_step / 1e3) / 1e9, 2),
                "pipeline_frac": round(alg_bytes / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBPS, 5),
                "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
                "kernel_sum_ms_per_step": round(total_ms / args.steps, 4),
            }
        # ---- the same frames with sub-node prediction switched OFF -----------
        # (blocks of a level are then independent: no dependency chain)
        if not args.no_extras and args.direction == "both":
            p1 = p.copy()
            p1.raht_subnode_prediction_enabled_flag = 0 if args.subnode else 1
            def step1():
                d_attrs.copy_(src)
                ctx.dev_raht_forward(p1, offsets, d_morton.data_ptr(), d_attrs.data_ptr(), d_coeffs.data_ptr(), c)
                ctx.dev_raht_inverse(p1, offsets, d_morton.data_ptr(), d_dec.data_ptr(), d_coeffs.data_ptr(), c)
            step1()
            torch.cuda.synchronize(dev)
            k1 = 10 if args.subnode else 3
            t1 = time.perf_counter()
            for _ in range(k1):
                step1()
            torch.cuda.synchronize(dev)
            dt1 = (time.perf_counter() - t1) / k1
            out["alt_flags"] = {
                "raht_subnode_prediction": int(p1.raht_subnode_prediction_enabled_flag),
                "value": round(n / dt1 / 1e6, 3), "unit": "Mpoints/s",
                "ms_per_step": round(dt1 * 1e3, 3), "steps": k1,
                "roundtrip_decoder_equals_encoder_recon": bool(torch.equal(d_attrs, d_dec))}
        if not args.no_extras:
            out["lifting"] = lifting_leg(ctx, args)
        # ---- CPU baseline: the compiled reference on one host core --------
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames[0], p, c)

    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(args, kernel, field=None):
    """HBM bytes per launch of the dominant kernel (FETCH_SIZE + WRITE_SIZE),
    from the committed rocprofv3 PMC passes of this exact workload
    (profiles/r01_pmc_traffic.json); None for any other workload.  `field`
    returns another recorded value instead, e.g. the kernel's average launch
    duration when the counters were collected (PMC passes cannot run inside
    this process: compare it with avg_launch_us to see how current they are)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    default = (args.cloud == "lidar" and args.points == 1_000_000 and args.frames == 1 and args.subnode == 1
               and args.qp == 34 and not args.haar and args.direction == "both")
    if not default or not os.path.exists(path):
        return None
    rec = json.load(open(path)).get(kernel)
    if rec is None:
        return None
    if field:
        return rec.get(field)
    return rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]


def lifting_leg(ctx, args):
    """BASELINE configs[2] shape on one GPU: the lifting attribute coder of one
    dense colour cloud minus the entropy loop -- encoder side = LoD build (kNN
    predictor search) + lifting forward, decoder side = LoD build + lifting
    inverse, each ONE call whose predictors stay on the device
    (gpcc_lift_encode_attr / gpcc_lift_decode_attr).  Host tier: positions and
    attributes come from and go to host buffers, so these rates include the
    PCIe copies."""
    from mpeg_pcc_tmc13_amd import lift_params, lod_params, synth
    n = min(args.points, 1_000_000)
    xyz, col = synth.dense_cloud(n, seed=77, bits=10)
    n = len(xyz)
    lp = lod_params()
    ctx.lift_encode_attr(lp, lift_params([1000], qp=34), xyz[:1000], col[:1000])  # warm-up (module load, arena)
    ctx.lift_encode_attr(lp, lift_params([n], qp=34), xyz, col)
    lf = lift_params([n], qp=34)
    t0 = time.perf_counter()
    co, rec, lcp, idx = ctx.lift_encode_attr(lp, lf, xyz, col)
    t_enc = time.perf_counter() - t0
    lf2 = lift_params([n], qp=34)
    t0 = time.perf_counter()
    dec = ctx.lift_decode_attr(lp, lf2, xyz, co, lcp)
    t_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    g = ctx.lod_build(lp, xyz)
    t_lod = time.perf_counter() - t0
    res = {"workload": f"{n}-point S-dense colour cloud, {lf.num_lods} LoDs, distance sub-sampling, 3 neighbours, qp 34",
           "encode_ms": round(t_enc * 1e3, 2), "decode_ms": round(t_dec * 1e3, 2),
           "lod_build_alone_ms": round(t_lod * 1e3, 2),
           "value": round(n / (t_enc + t_dec) / 1e6, 3), "unit": "Mpoints/s (encode + decode, host buffers, PCIe inclusive)",
           "roundtrip_decoder_equals_encoder_recon": bool(np.array_equal(np.asarray(dec), np.asarray(rec)))}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import lod_helpers as lh
        import oracle_loader as ol
        kind = "reference" if ol.ref_available() else "port"
        t0 = time.perf_counter()
        o = (lh.ref_lod_generate if kind == "reference" else lh.oracle_lod_generate)(xyz, lp)
        t_ref = time.perf_counter() - t0
        same = all(np.array_equal(np.asarray(g[k]).astype(np.int64), np.asarray(o[k]).astype(np.int64))
                   for k in ("npl", "indexes", "nc", "ni", "w"))
        res["cpu_lod_build"] = {"value": round(n / t_ref / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": kind,
                                "sample": f"AttributeLods::generate on the same {n} points, {t_ref:.2f} s "
                                          "(needed once by the encoder and once by the decoder)",
                                "gpu_result_identical": bool(same)}
    return res


def cpu_baseline(frame, p, c):
    """oracle/_ref (the reference's own RAHT.cpp, -O3) forward+inverse on the
    same frame, one core; falls back to the C oracle port if the compiled
    reference did not travel."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_loader as ol
    morton, attrs, _ = frame
    kind = "reference" if ol.ref_available() else "port"
    chk = ol.ref() if kind == "reference" else ol.oracle()
    n = len(morton)
    # bounded sample: the whole frame, repeated until about 10 s of CPU work
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while reps < 12 and dt < 10.0:
        co, rec = chk.raht_forward(p, morton, attrs)
        chk.raht_inverse(p, morton, co, c)
        reps += 1
        dt = time.perf_counter() - t0
    return {"value": round(n * reps / dt / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": kind,
            "sample": f"forward+inverse of frame 0 ({n} points, same flags) x {reps}, "
                      f"{dt:.1f} s wall, host {os.cpu_count()} logical cores, 1 used"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- RAHT attribute-transform throughput on MI355X.

One STEP = one pass of the hot path over one batch of synthetic,
Morton-sorted frames resident in HBM: RAHT forward (encoder side:
coefficients + reconstruction) followed by RAHT inverse (decoder side) of
every frame of the batch.  At N=1 the workload is BASELINE.json configs[1]:
a 1M-point lidar-shaped cloud (S-lidar, 18-bit grid, reflectance, C=1),
flags of cfg/octree-raht-ctc-lossless-geom-lossy-attrs.yaml (qp 34, search
range 2500) and the reference's defaults for everything else, i.e.
raht_subnode_prediction_enabled_flag = 1 (TMC3.cpp:1307).

Extra objects of the same JSON line (rank 0, N=1):
  roofline          dominant kernel of the headline step, priced with the
                    algorithmic bytes of ITS direction (SURVEY.md 8(d):
                    forward 8+12C, inverse 8+8C bytes per point)
  raht_forward_10M  the north-star target configuration: RAHT FORWARD of 10
                    slices x 1M points in one batch, both states of the
                    sub-node prediction flag, with its own roofline
  alt_flags         the headline frames with the sub-node flag flipped
  hbm_calibration   a device copy on this box next to the nominal 8 TB/s
  lifting           LoD build + lifting (configs[2] shape)
  cpu_baseline      the compiled reference on one host core and on all of
                    them (one process per frame)

With N>1 every rank transforms its own frame(s) (weak scaling: per-GPU work
is fixed, frames shard one-per-GPU) and the quantised coefficients are
gathered on rank 0 with one RCCL gather inside the timed region; the line
then also carries `configs3`, the same measurement on BASELINE configs[3]'s
2M-point dense frames.  What scales is transform + gather: the arithmetic
coder that consumes the coefficients is the reference's CPU code and is not
in the loop.

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

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured with a float4 copy)


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
    ap.add_argument("--direction", choices=["both", "forward", "inverse"], default="both",
                    help="inverse: decoder only (coefficients prepared on the CPU outside the timed region)")
    ap.add_argument("--configs3-points", type=int, default=2_000_000,
                    help="points per frame of the N>1 configs[3] leg")
    ap.add_argument("--verify-gather", dest="verify_gather", action="store_true", default=True,
                    help="N>1 (default): rank 0 recomputes every rank's frames itself and compares with what it gathered")
    ap.add_argument("--no-verify-gather", dest="verify_gather", action="store_false")
    ap.add_argument("--configs4-points", type=int, default=1_000_000,
                    help="points per slice of the N>1 configs[4] leg (ten slices, colour + reflectance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--legs", default="",
                    help="N=1: run ONLY these extra legs (comma separated: lifting,predicting,recolour,raht_inter) after "
                         "the headline step -- for profiler passes of one leg (tools/pmc.sh)")
    ap.add_argument("--frames-per-gpu-batched", type=int, default=10,
                    help="frames per GPU of the weak_batched leg (the regime where one GPU is busy)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the alternative-flag, 10M-forward, calibration and lifting legs")
    return ap.parse_args()


def make_frame(cloud, points, seed, refl_noise=6):
    from mpeg_pcc_tmc13_amd import synth
    if cloud == "lidar":
        xyz, attrs = synth.lidar_cloud(points, seed=seed, refl_noise=refl_noise)
        bits = 18
    else:
        bits = 10 if points <= 1_500_000 else 12
        xyz, attrs = synth.dense_cloud(points, seed=seed, bits=bits)
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    return morton, attrs, 3 * bits


class Batch:
    """Frames of one rank, resident in HBM, and the step over them."""

    def __init__(self, torch, dev, ctx, frames, params):
        self.torch, self.ctx, self.p = torch, ctx, params
        self.c = frames[0][1].shape[1]
        sizes = [len(f[0]) for f in frames]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.n = int(self.offsets[-1])
        self.bits = frames[0][2]
        self.d_morton = to_device(torch, np.concatenate([f[0] for f in frames]), dev)
        self.src = to_device(torch, np.concatenate([f[1] for f in frames]).reshape(-1), dev)
        self.d_attrs = torch.empty_like(self.src)
        self.d_coeffs = torch.zeros(self.c * self.n, dtype=torch.int32, device=dev)
        self.d_dec = torch.empty_like(self.src)

    def forward(self, p=None):
        self.d_attrs.copy_(self.src)  # the transform overwrites its input with the reconstruction
        self.ctx.set_morton_bits(self.bits)
        self.ctx.dev_raht_forward(p or self.p, self.offsets, self.d_morton.data_ptr(),
                                  self.d_attrs.data_ptr(), self.d_coeffs.data_ptr(), self.c)

    def inverse(self, p=None):
        self.ctx.set_morton_bits(self.bits)
        self.ctx.dev_raht_inverse(p or self.p, self.offsets, self.d_morton.data_ptr(),
                                  self.d_dec.data_ptr(), self.d_coeffs.data_ptr(), self.c)

    def roundtrip_ok(self):
        return bool(self.torch.equal(self.d_attrs, self.d_dec))

    def bytes_forward(self):
        return self.n * (8 + 12 * self.c)

    def bytes_inverse(self):
        return self.n * (8 + 8 * self.c)


def timed(torch, dev, fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def timed_stats(torch, dev, fn, steps, warmup=2, settle=0.0):
    """every step timed on its own (a synchronisation per step): (median, max, index of the slowest step) tell a
    steady state from a hiccup (arena regrowth, an expired bounded wait and its retry).
    `settle`: an untimed pause between the warm-up calls and the timed steps.  On the MI355X box the first one or two
    calls right behind a large ALLOCATION (a batch's buffers from torch's allocator, the context's arena regrown by
    the first warm-up call) took 10-80 ms with every kernel at its usual time (profiles/r05_stall_root_cause.txt);
    round 5 benchmarked around it with a 0.3 s pause.  Since round 6 the library has gpcc_ctx_reserve -- the bench
    reserves once, before any batch exists -- and the legs are timed with settle = 0; the batch curve reports both
    variants so that the pause's effect stays visible."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    if settle:
        time.sleep(settle)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    slowest = max(range(len(ts)), key=lambda i: ts[i])
    ts.sort()
    return ts[len(ts) // 2], ts[-1], slowest


def kernel_profile(torch, dev, ctx, fn, steps):
    """{kernel: (ms per step, launches per step)}: HIP events on the context's stream."""
    ctx.set_profiling(True)
    ctx.kernel_times()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    kt = ctx.kernel_times()
    ctx.set_profiling(False)
    return {k: (v[0] / steps, v[1] / steps) for k, v in kt.items()}


def to_device(torch, array, dev):
    """an upload that starts from pinned memory: out of pageable memory the ROCm runtime pins the source pages on the
    fly, read-only, and keeps the pins -- a later download onto reused heap addresses then faults (the library's own
    transfers go through a pinned bounce buffer for the same reason, csrc/gpcc_attr_mi355.hip h2d_user)"""
    t = torch.from_numpy(np.ascontiguousarray(array))
    if dev.type == "cuda" and t.numel():
        t = t.pin_memory()
    return t.to(dev)


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
    dist_info = None
    if world > 1:
        # what the collective really is: checked AFTER init, reported in the line
        assert dist.is_initialized() and dist.get_world_size() == args.gpus, \
            f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}"
        assert dist.get_backend() == backend, (dist.get_backend(), backend)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "collective_library": ("RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version()))
                     if backend == "nccl" else backend,
                     "devices_visible_per_rank": torch.cuda.device_count(),
                     "hip": getattr(torch.version, "hip", None)}
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where collectives operate

    def params_for(cloud, subnode, haar=False, qp=args.qp):
        if haar:
            return raht_params(qp=4, haar=True, chroma_offset=0, subnode=bool(subnode), search_range=2500)
        return raht_params(qp=qp, subnode=bool(subnode), search_range=2500 if cloud == "lidar" else 50000)

    p = params_for(args.cloud, args.subnode, bool(args.haar))

    # One non-default stream carries everything -- torch's copies, the
    # library's kernels, the collective -- so the step is ordered by the
    # stream, not by host synchronisation.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(local_rank, stream=stream.cuda_stream)

    frames = [make_frame(args.cloud, args.points, seed=1 + rank * args.frames + f) for f in range(args.frames)]
    # everything the transforms allocate on demand, once, before any batch exists (gpcc_ctx_reserve): the legs below
    # are timed without a pause behind their warm-up (timed_stats)
    ctx.set_morton_bits(frames[0][2])
    biggest = max(args.points * args.frames, 0 if args.no_extras else max(
        10 * 1_000_000, 10 * args.configs4_points, args.frames_per_gpu_batched * args.points, args.configs3_points))
    ctx.reserve(biggest, max(args.frames, 10), 3)
    b = Batch(torch, dev, ctx, frames, p)
    c, n = b.c, b.n

    decoder_input_check = None
    if args.direction == "inverse":
        # decoder-only run: the coefficients come from the device's own encoder, once, outside the timed region
        b.forward()
        torch.cuda.synchronize(dev)
        # ... and ONE untimed comparison of that encoder with the CPU (the compiled reference where it is built,
        # else the oracle): a defect shared by the device encoder and decoder would otherwise still report
        # roundtrip_ok
        if rank == 0:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_loader as ol
                cpu = ol.ref() if ol.ref_available() else ol.oracle()
                m0, a0, _ = frames[0]
                co_cpu, rec_cpu = cpu.raht_forward(p, m0, a0)
                n0 = len(m0)
                co_dev = b.d_coeffs[:n0 * c].cpu().numpy()
                rec_dev = b.d_attrs[:n0 * c].cpu().numpy()
                decoder_input_check = {
                    "cpu": "reference" if ol.ref_available() else "oracle", "points": n0,
                    "coefficients_equal": bool(np.array_equal(co_dev, np.asarray(co_cpu).reshape(-1))),
                    "reconstruction_equal": bool(np.array_equal(rec_dev, np.asarray(rec_cpu).reshape(-1)))}
            except Exception as e:  # (no checker on this box: say so in the line)
                decoder_input_check = {"cpu": None, "error": repr(e)[:200]}

    def gather_step(batch, gathered):
        if world > 1:
            dist.gather(batch.d_coeffs if backend == "nccl" else batch.d_coeffs.cpu(), gathered, dst=0)

    def make_gather_buffers(batch):
        return ([torch.empty_like(batch.d_coeffs, device=xdev) for _ in range(world)]
                if (world > 1 and rank == 0) else None)

    def run_timed(batch, direction, steps, warmup):
        gathered = make_gather_buffers(batch)

        def step():
            if direction in ("both", "forward"):
                batch.forward()
            if direction in ("both", "inverse"):
                batch.inverse()
            gather_step(batch, gathered)

        def fence():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        ctx.synchronize()  # raises if any launch of the loop reported a device-side error
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, gathered

    elapsed, gathered = run_timed(b, args.direction, args.steps, args.warmup)
    # in-run sanity: decoder output == encoder reconstruction (the
    # reference's own conformance criterion)
    roundtrip_ok = b.roundtrip_ok() if args.direction != "forward" else None
    ms_per_step = elapsed / args.steps * 1e3
    value = n * world * args.steps / elapsed / 1e6
    dir_name = {"both": "forward+inverse", "forward": "forward only", "inverse": "inverse only"}[args.direction]

    out = {
        "metric": "RAHT forward+inverse attribute-transform Mpoints/s (bit-exact vs CPU reference)",
        "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int64 (Q15 fixed point), int32 I/O",
        "data": "synthetic",
        "config": {
            # (flag state first, <= 120 characters: the driver's parsed copy keeps 128)
            "workload": f"subnode={int(args.subnode)} pred=1 ext=1 "
                        + ("haar qp4" if args.haar else f"qp{args.qp}")
                        + f": RAHT {dir_name}, {args.frames}x{args.points} Morton-sorted S-{args.cloud}/GPU, C={c}",
            "points_per_gpu_per_step": n, "frames_per_gpu": args.frames,
            "roundtrip_decoder_equals_encoder_recon": roundtrip_ok,
        },
    }
    if decoder_input_check is not None:
        out["config"]["decoder_input_vs_cpu_encoder"] = decoder_input_check
    if world > 1:
        out["distributed"] = dist_info
        out["config"]["scaling_scope"] = ("transform + one RCCL gather of the coefficient buffers; the CPU "
                                          "arithmetic coder that consumes them is not in the loop")

    if world > 1:
        # where a step's time goes on every rank: the transforms alone (device time, no collective) and the
        # gather alone, 3 steps each, every rank's figure on rank 0
        def rank_times(batch):
            gathered_ = make_gather_buffers(batch)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                if args.direction in ("both", "forward"):
                    batch.forward()
                if args.direction in ("both", "inverse"):
                    batch.inverse()
            torch.cuda.synchronize(dev)
            dev_ms = (time.perf_counter() - t0) / 3 * 1e3
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                gather_step(batch, gathered_)
            torch.cuda.synchronize(dev)
            dist.barrier()
            gat_ms = (time.perf_counter() - t0) / 3 * 1e3
            mine_ = torch.tensor([dev_ms, gat_ms], dtype=torch.float64, device=xdev)
            allr = [torch.zeros(2, dtype=torch.float64, device=xdev) for _ in range(world)] if rank == 0 else None
            dist.gather(mine_, allr, dst=0)
            if rank != 0:
                return None
            return {"device_ms_per_rank": [round(float(t[0]), 3) for t in allr],
                    "gather_ms_per_rank": [round(float(t[1]), 3) for t in allr]}
        rt = rank_times(b)
        if rank == 0:
            out["distributed"].update(rt)

    if not args.no_extras and args.direction == "both":
        # ---- the weak-scaling line in the regime where ONE GPU is busy: frames_per_gpu_batched frames per GPU and
        #      step (a single 1 M-point frame is a latency-bound unit: its time says nothing about throughput).
        #      The same leg runs at N = 1, so the N-GPU figure divides by a like-for-like single-GPU figure. ----
        fb = [make_frame(args.cloud, args.points, seed=1000 + rank * args.frames_per_gpu_batched + f)
              for f in range(args.frames_per_gpu_batched)]
        bb = Batch(torch, dev, ctx, fb, p)
        kb = 3
        elb, _ = run_timed(bb, "both", kb, 1)
        wb = {"workload": f"{args.frames_per_gpu_batched} x {args.points}-point S-{args.cloud} frames per GPU and step, "
                          "forward+inverse, default flags" + (" + gather of the coefficient buffers" if world > 1 else ""),
              "frames_per_gpu": args.frames_per_gpu_batched, "scaling": "weak",
              "value": round(bb.n * world * kb / elb / 1e6, 3), "unit": "Mpoints/s",
              "ms_per_step": round(elb / kb * 1e3, 3), "steps": kb,
              "roundtrip_decoder_equals_encoder_recon": bb.roundtrip_ok()}
        if world > 1:
            rtb = rank_times(bb)
            if rank == 0:
                wb.update(rtb)
        if rank == 0:
            out["weak_batched"] = wb
        del bb

    if world > 1 and args.verify_gather and rank == 0 and args.direction != "inverse":
        # what rank 0 gathered is what a single GPU computes for the same frames
        same = True
        for r in range(world):
            fr = [make_frame(args.cloud, args.points, seed=1 + r * args.frames + f) for f in range(args.frames)]
            br = Batch(torch, dev, ctx, fr, p)
            br.forward()
            torch.cuda.synchronize(dev)
            same = same and bool(torch.equal(br.d_coeffs.to(xdev), gathered[r]))
            del br
        out["config"]["gathered_equals_single_rank"] = same

    if world > 1 and not args.no_extras:
        # ---- BASELINE configs[3]: 2M-point dense frames, one per GPU ------
        f3 = [make_frame("dense", args.configs3_points, seed=101 + rank)]
        b3 = Batch(torch, dev, ctx, f3, params_for("dense", 1))
        k3 = 5
        el3, _ = run_timed(b3, "both", k3, 1)
        npts = torch.tensor([b3.n], dtype=torch.float64, device=xdev)
        dist.all_reduce(npts, op=dist.ReduceOp.SUM)
        if rank == 0:
            out["configs3"] = {
                "workload": f"{world} x {args.configs3_points}-point S-dense colour frames (C=3, default flags), one per GPU, "
                            "forward+inverse + RCCL gather of the coefficients",
                "value": round(float(npts.item()) * k3 / el3 / 1e6, 3), "unit": "Mpoints/s",
                "ms_per_step": round(el3 / k3 * 1e3, 3), "steps": k3,
                "roundtrip_decoder_equals_encoder_recon": b3.roundtrip_ok()}
        del b3

    if not args.no_extras and args.direction == "both":
        c4 = configs4_leg(torch, dist, dev, xdev, ctx, args, params_for, world, rank, backend)
        if rank == 0:
            out["configs4"] = c4

    if rank == 0 and world == 1:
        if not args.no_profile:
            out["roofline"] = roofline(torch, dev, ctx, b, args)
        if not args.no_extras and args.direction == "both":
            # ---- the same frames with the sub-node flag flipped ------------
            p1 = params_for(args.cloud, 0 if args.subnode else 1, bool(args.haar))

            def step1():
                b.forward(p1)
                b.inverse(p1)
            k1 = 10
            dt1 = timed(torch, dev, step1, k1)
            out["alt_flags"] = {
                "raht_subnode_prediction": int(p1.raht_subnode_prediction_enabled_flag),
                "value": round(n / dt1 / 1e6, 3), "unit": "Mpoints/s",
                "ms_per_step": round(dt1 * 1e3, 3), "steps": k1,
                "roundtrip_decoder_equals_encoder_recon": b.roundtrip_ok()}
        if not args.no_extras and args.direction == "both" and not args.haar:
            # ---- the headline frame where the entropy is not trivial: at qp 34 the smooth S-lidar
            #      reflectance codes to a 907-byte payload (nearly every coefficient zero: neither the
            #      undecided band of the RDOQ nor the zero-run / binarisation front end carry load);
            #      qp 22 and qp 10 put most coefficients through them.  Cost across qp, same kernels. ---
            out["qp_sweep"] = {}
            for q in (22, 10):
                pq = params_for(args.cloud, args.subnode, False, qp=q)

                def stepq():
                    b.forward(pq)
                    b.inverse(pq)
                dtq = timed(torch, dev, stepq, 5)
                b.forward(pq)
                torch.cuda.synchronize(dev)
                co = b.d_coeffs
                nz = float((co != 0).sum().item()) / co.numel()
                # the entropy front end on the device: zero runs + binarisation of this frame's coefficients
                # (host tier: PCIe both ways included; the second of two calls -- the first grows the pool)
                h_co = co.cpu().numpy()
                for _ in range(2):
                    t0 = time.perf_counter()
                    runs, syms, trailing = ctx.zero_run_pack(h_co, n, c, planar=True)
                    bins = ctx.binarise_symbols(runs, syms, trailing, c)
                    front_ms = (time.perf_counter() - t0) * 1e3
                b.inverse(pq)
                out["qp_sweep"][f"qp{q}"] = {
                    "value": round(n / dtq / 1e6, 3), "unit": "Mpoints/s", "ms_per_step": round(dtq * 1e3, 3),
                    "nonzero_coefficient_fraction": round(nz, 4), "symbols": int(len(runs)), "bins": int(len(bins)),
                    "entropy_front_end_ms_host_tier": round(front_ms, 2),
                    "roundtrip_decoder_equals_encoder_recon": b.roundtrip_ok()}
        if not args.no_extras and args.direction == "both" and args.cloud == "lidar" and not args.haar:
            # ---- the headline workload with a TEXTURED reflectance field (noise +-24 instead of +-6 on the same
            #      scene): at qp 34 several per cent of the coefficients survive, so the RDOQ chain of the
            #      lossy encoder is in the number (the smooth headline frame codes to 0.1 % non-zero) ----------
            ft = [make_frame("lidar", args.points, seed=1, refl_noise=24)]
            bt = Batch(torch, dev, ctx, ft, p)

            def stept():
                bt.forward()
                bt.inverse()
            med, mx, _ = timed_stats(torch, dev, stept, 10, warmup=2)
            bt.forward()
            torch.cuda.synchronize(dev)
            nzt = float((bt.d_coeffs != 0).sum().item()) / bt.d_coeffs.numel()
            bt.inverse()
            out["headline_textured"] = {
                "workload": f"the headline flags on a {args.points}-point S-lidar frame with reflectance noise +-24",
                "value": round(bt.n / med / 1e6, 3), "unit": "Mpoints/s", "ms_per_step": round(med * 1e3, 3),
                "ms_per_step_max": round(mx * 1e3, 3), "steps": 10,
                "nonzero_coefficient_fraction": round(nzt, 4),
                "roundtrip_decoder_equals_encoder_recon": bt.roundtrip_ok()}
            del bt
            # (next to `value`: the headline cannot be read without it -- VERDICT r05)
            out["config"]["same_call_on_textured_field"] = {
                "value": out["headline_textured"]["value"], "unit": "Mpoints/s",
                "ms_per_step": out["headline_textured"]["ms_per_step"],
                "nonzero_coefficient_fraction": out["headline_textured"]["nonzero_coefficient_fraction"]}
        if args.legs:
            for leg in args.legs.split(","):
                out[leg] = {"first_call": lambda: first_call_leg(torch, dev, local_rank, stream, frames[0], p),
                            "lifting": lambda: lifting_leg(ctx, args), "predicting": lambda: predicting_leg(ctx, args),
                            "recolour": lambda: recolour_leg(ctx, args), "raht_inter": lambda: raht_inter_leg(ctx, args)}[leg]()
        if not args.no_extras:
            out["raht_forward_10M"] = forward_10m(torch, dev, ctx, params_for, frames)
            out["hbm_calibration"] = hbm_calibration(torch, dev)
            # the same fractions against what a plain copy reaches on THIS box
            cal = out["hbm_calibration"]["copy_GBps"]
            rl = [out.get("roofline")] + [out["raht_forward_10M"].get(k, {}).get("roofline") for k in ("subnode_0", "subnode_1")]
            for r in rl:
                if r:
                    r["frac_calibrated"] = round(r["achieved"] / cal, 5)
                    r["pipeline_frac_calibrated"] = round(r["pipeline_achieved"] / cal, 5)
            out["first_call"] = first_call_leg(torch, dev, local_rank, stream, frames[0], p)
            out["host_tier"] = host_tier_leg(ctx, frames[0], p)
            out["lifting"] = lifting_leg(ctx, args)
            out["predicting"] = predicting_leg(ctx, args)
            out["recolour"] = recolour_leg(ctx, args)
            out["raht_inter"] = raht_inter_leg(ctx, args)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames[0], p, c)

    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def configs4_leg(torch, dist, dev, xdev, ctx, args, params_for, world, rank, backend):
    """BASELINE configs[4]'s shape as a STRONG-scaling leg: a 10M-point frame = ten 1M-point slices
    (tmc3/encoder.cpp:544-571 codes slice after slice) with colour (C=3) AND reflectance (C=1) on the same
    geometry, the slices sharded over the ranks by sharding.shard_units (longest first), every rank transforms
    colour then reflectance of its slices (forward + inverse, reference default flags) and the coefficient
    buffers of both attributes are gathered on rank 0.  Total work is fixed, so with ten equal slices on
    eight ranks (2,2,1,1,1,1,1,1) the speed-up over one GPU is at most 10 / 2 = 5x BY CONSTRUCTION --
    `ceiling` states it; the weak-scaling headline is the line the >= 6x target is read from."""
    from mpeg_pcc_tmc13_amd import sharding, synth
    n_slices, pts = 10, args.configs4_points
    assign = sharding.shard_units([pts] * n_slices, world)
    mine = assign[rank]
    frames_c, frames_r = [], []
    for u in mine:
        xyz, col = synth.dense_cloud(pts, seed=401 + u, bits=10 if pts <= 1_500_000 else 12)
        rng = np.random.default_rng(9000 + u)
        refl = np.clip(col[:, :1] * 3 // 4 + rng.integers(-6, 7, size=(len(col), 1)), 0, 255).astype(np.int32)
        m, a, order = synth.sort_by_morton(xyz, col)
        frames_c.append((m, a, 30 if pts <= 1_500_000 else 36))
        frames_r.append((m, np.ascontiguousarray(refl[order]), frames_c[-1][2]))
    p = params_for("dense", 1)
    bc = Batch(torch, dev, ctx, frames_c, p) if mine else None
    br = Batch(torch, dev, ctx, frames_r, p) if mine else None
    my_n = bc.n if mine else 0
    # one padded buffer per rank: colour coefficients, then reflectance coefficients
    ln = torch.tensor([4 * my_n], dtype=torch.int64, device=xdev)
    if world > 1:
        dist.all_reduce(ln, op=dist.ReduceOp.MAX)
    cap = int(ln.item())
    send = torch.zeros(cap, dtype=torch.int32, device=dev)
    gathered = [torch.empty(cap, dtype=torch.int32, device=xdev) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        if mine:
            bc.forward()
            bc.inverse()
            br.forward()
            br.inverse()
            send[:3 * my_n].copy_(bc.d_coeffs)
            send[3 * my_n:4 * my_n].copy_(br.d_coeffs)
        if world > 1:
            dist.gather(send if backend == "nccl" else send.cpu(), gathered, dst=0)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    k = 3
    step()
    fence()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    fence()
    el = time.perf_counter() - t0
    ctx.synchronize()
    ok = (bc.roundtrip_ok() and br.roundtrip_ok()) if mine else True
    tot = torch.tensor([float(my_n), 1.0 if ok else 0.0], dtype=torch.float64, device=xdev)
    tmax = torch.tensor([el], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())
    same = None
    if world > 1 and rank == 0 and args.verify_gather:
        # what arrived from every other rank is what this GPU computes for that rank's slices
        same = True
        for r in range(1, world):
            off = 0
            for u in assign[r]:
                xyz, col = synth.dense_cloud(pts, seed=401 + u, bits=10 if pts <= 1_500_000 else 12)
                m, a, _ = synth.sort_by_morton(xyz, col)
                b1 = Batch(torch, dev, ctx, [(m, a, 30 if pts <= 1_500_000 else 36)], p)
                b1.forward()
                torch.cuda.synchronize(dev)
                # (rank r's colour block is laid out slice after slice, planar per slice)
                same = same and bool(torch.equal(b1.d_coeffs.to(xdev), gathered[r][off:off + 3 * b1.n]))
                off += 3 * b1.n
                del b1
    per_rank = [len(a) for a in assign]
    res = {
        "workload": f"{n_slices} x {pts}-point S-dense slices, colour (C=3) then reflectance (C=1) on the same geometry, "
                    f"forward+inverse, default flags, LPT-sharded over {world} rank(s) + gather of both coefficient sets",
        "scaling": "strong", "slices_per_rank": per_rank,
        "ceiling": f"at most {n_slices / max(per_rank):.2f}x one GPU by construction ({max(per_rank)} slices on the fullest rank)",
        "value": round(float(tot[0].item()) * k / el / 1e6, 3), "unit": "Mpoints/s (points of the frame, both attributes coded)",
        "ms_per_step": round(el / k * 1e3, 3), "steps": k,
        "roundtrip_decoder_equals_encoder_recon": bool(tot[1].item() == world),
    }
    if same is not None:
        res["gathered_equals_single_rank"] = same
    return res


def roofline(torch, dev, ctx, b, args):
    """Dominant kernel of the headline step.  Forward and inverse are profiled
    separately (HIP events on the context's stream), so every kernel is priced
    with the algorithmic bytes of its own direction; one launch is credited
    with that direction's bytes / its launches per pass."""
    steps = max(3, min(args.steps, 10))
    prof = {}
    if args.direction in ("both", "forward"):
        prof["forward"] = (kernel_profile(torch, dev, ctx, b.forward, steps), b.bytes_forward())
    if args.direction in ("both", "inverse"):
        if args.direction == "inverse":
            pass
        else:
            b.forward()
        prof["inverse"] = (kernel_profile(torch, dev, ctx, b.inverse, steps), b.bytes_inverse())
    dom = None
    for direction, (kt, nbytes) in prof.items():
        for k, (ms, launches) in kt.items():
            if dom is None or ms > dom[2]:
                dom = (direction, k, ms, launches, nbytes)
    direction, name, ms, launches, nbytes = dom
    achieved = nbytes / (ms / 1e3) / 1e9
    total_ms = sum(ms_ for kt, _ in prof.values() for ms_, _ in kt.values())
    total_bytes = sum(nb for _, nb in prof.values())
    res = {
        "bound": "hbm", "kernel": name, "direction": direction,
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 5),
        "traffic": pmc_traffic(args, name),
        "launches_per_step": round(launches, 2),
        "algorithmic_bytes_per_launch": round(nbytes / launches),
        "avg_launch_us": round(ms * 1e3 / launches, 2),
        "pipeline_achieved": round(total_bytes / (total_ms / 1e3) / 1e9, 2),
        "pipeline_frac": round(total_bytes / (total_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 5),
    }
    for direction, (kt, nbytes) in prof.items():
        res[f"{direction}_kernel_ms"] = {k: round(v[0], 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}
        res[f"{direction}_launches"] = round(sum(v[1] for v in kt.values()), 1)
    return res


def forward_10m(torch, dev, ctx, params_for, first_frames):
    """The north-star target configuration: RAHT FORWARD of 10 slices of 1M
    points (a 10M-point frame as the reference would slice it) in
    one batch, inputs resident in HBM; reference default flags and the
    sub-node flag off."""
    frames = list(first_frames[:1]) if len(first_frames[0][0]) >= 900_000 and first_frames[0][1].shape[1] == 1 else []
    while len(frames) < 10:
        frames.append(make_frame("lidar", 1_000_000, seed=1 + len(frames)))
    res = {"workload": "RAHT forward (coefficients + reconstruction), 10 x 1M-point S-lidar slices in one batch, "
                       "C=1, qp 34, search range 2500", "bytes_per_point": 20}
    for sub in (1, 0):
        p = params_for("lidar", sub)
        b = Batch(torch, dev, ctx, frames, p)
        # (median of 10 steps timed one by one: one step in a few hundred lands in the tens of milliseconds the
        # device's queue can idle behind a large free / allocation -- timed_stats -- and a mean of five would
        # report that instead of the transform; the slowest step is in the line)
        k = 10
        dt, dt_max, dt_at = timed_stats(torch, dev, b.forward, k, warmup=2)
        ctx.synchronize()
        kt = kernel_profile(torch, dev, ctx, b.forward, 3)
        name, (ms, launches) = max(kt.items(), key=lambda kv: kv[1][0])
        nbytes = b.bytes_forward()
        achieved = nbytes / (ms / 1e3) / 1e9
        res[f"subnode_{sub}"] = {
            "value": round(b.n / dt / 1e6, 2), "unit": "Mpoints/s", "ms_per_forward": round(dt * 1e3, 3),
            "ms_per_forward_max": round(dt_max * 1e3, 3), "slowest_step": dt_at,
            "steps": k, "launches_per_forward": round(sum(v[1] for v in kt.values()), 1),
            "roofline": {
                "bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": pmc_traffic(None, name, f"fwd10_sub{sub}"),
                "launches_per_forward": round(launches, 1),
                "algorithmic_bytes_per_launch": round(nbytes / launches),
                "avg_launch_us": round(ms * 1e3 / launches, 2),
                "pipeline_achieved": round(nbytes / dt / 1e9, 2),
                "pipeline_frac": round(nbytes / dt / 1e9 / HBM_PEAK_GBPS, 5)},
            "kernel_ms": {k_: round(v[0], 4) for k_, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
        }
        del b
    # throughput against the number of slices in flight (the
    # per-frame dependency chains overlap across slices; the asymptote is what a
    # node full of slices sees); per step: median of 10, and the slowest of them
    curve = {}
    for sub in (1, 0):
        p = params_for("lidar", sub)
        pts = []
        for nf in (1, 2, 5, 10):
            b = Batch(torch, dev, ctx, frames[:nf], p)
            tf, tf_max, tf_at = timed_stats(torch, dev, b.forward, 10, warmup=2)
            ti, ti_max, ti_at = timed_stats(torch, dev, b.inverse, 10, warmup=2)
            # the same with round 5's 0.3 s pause behind the warm-up (see timed_stats)
            sf, sf_max, _ = timed_stats(torch, dev, b.forward, 10, warmup=1, settle=0.3)
            si, si_max, _ = timed_stats(torch, dev, b.inverse, 10, warmup=1, settle=0.3)
            pts.append({"slices": nf, "steps": 10, "forward_ms": round(tf * 1e3, 3), "forward_ms_max": round(tf_max * 1e3, 3),
                        "forward_slowest_step": tf_at, "inverse_slowest_step": ti_at,
                        "inverse_ms": round(ti * 1e3, 3), "inverse_ms_max": round(ti_max * 1e3, 3),
                        "settled_0p3s": {"forward_ms": round(sf * 1e3, 3), "forward_ms_max": round(sf_max * 1e3, 3),
                                         "inverse_ms": round(si * 1e3, 3), "inverse_ms_max": round(si_max * 1e3, 3)},
                        "forward_Mpts": round(b.n / tf / 1e6, 1), "inverse_Mpts": round(b.n / ti / 1e6, 1)})
            del b
        curve[f"subnode_{sub}"] = pts
    res["batch_curve"] = curve
    return res


def first_call_leg(torch, dev, local_rank, stream, frame, p, trials=20):
    """The reference's call pattern through the seams (tmc3/AttributeEncoder.cpp:1273, 1341): a context, then ONE
    transform per (slice, attribute).  Per trial: a FRESH context, the first forward transform of the headline frame
    timed on its own, no pause anywhere -- with gpcc_ctx_reserve in front (what INTEGRATION.md tells a seam user to
    do) and without it (the first call then pays for the arena).  max / median of the reserved variant is the
    figure: a call right behind the allocation used to take 10-80 ms now and then (profiles/r05_stall_root_cause.txt)."""
    from mpeg_pcc_tmc13_amd import context
    morton, attrs, bits = frame
    d_m = to_device(torch, morton, dev)
    src = to_device(torch, attrs.reshape(-1), dev)
    c = attrs.shape[1]
    n = len(morton)
    offs = np.array([0, n], dtype=np.int64)
    res = {}
    for name, reserve in (("reserved", True), ("unreserved", False)):
        first, second = [], []
        for _ in range(trials):
            cx = context(local_rank, stream=stream.cuda_stream)
            cx.set_morton_bits(bits)
            if reserve:
                cx.reserve(n, 1, c)
            d_a = src.clone()
            d_c = torch.zeros(c * n, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            for lst in (first, second):
                d_a.copy_(src)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                cx.dev_raht_forward(p, offs, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), c)
                torch.cuda.synchronize(dev)
                lst.append(time.perf_counter() - t0)
            cx.close()
            del d_a, d_c
        f = sorted(first)
        g = sorted(second)
        res[name] = {"trials": trials, "first_call_ms_median": round(f[len(f) // 2] * 1e3, 3),
                     "first_call_ms_max": round(f[-1] * 1e3, 3),
                     "max_over_median": round(f[-1] / f[len(f) // 2], 2),
                     "second_call_ms_median": round(g[len(g) // 2] * 1e3, 3), "second_call_ms_max": round(g[-1] * 1e3, 3)}
    res["note"] = "forward transform of the headline frame, device tier, one call per fresh context; no pause"
    return res


def host_tier_leg(ctx, frame, p):
    """What a caller of the reference's free functions sees (SURVEY.md 8(d) item i): the
    HOST tier on the headline frame -- pageable host buffers in, host buffers out, PCIe
    copies and the synchronisation included.  Never `value`."""
    morton, attrs, _ = frame
    ctx.raht_forward(p, morton, attrs)  # warm-up (pool, arena)
    t0 = time.perf_counter()
    co, rec = ctx.raht_forward(p, morton, attrs)
    t1 = time.perf_counter()
    ctx.raht_inverse(p, morton, co, attrs.shape[1])
    t2 = time.perf_counter()
    n = len(morton)
    return {"forward_ms": round((t1 - t0) * 1e3, 3), "inverse_ms": round((t2 - t1) * 1e3, 3),
            "Mpoints_per_s_fwd_inv": round(n / (t2 - t0) / 1e6, 2),
            "note": "gpcc_raht_forward / gpcc_raht_inverse on host buffers: H2D + transform + D2H, synchronous"}


def recolour_leg(ctx, args):
    """SURVEY.md 8(f) rank 1, the step upstream of the transforms in BASELINE configs[4]: the
    attributes of a 1M-point S-dense colour cloud transferred to its half-resolution geometry
    (pcc::recolour, the reference's defaults: 8 forward / 1 backward neighbours, +-1 refinement).
    Host tier (host buffers in and out).  Algorithmic bytes: source 12 + 4c, target 12 in, 4c out."""
    from mpeg_pcc_tmc13_amd import recolour_params, synth
    n = min(args.points, 1_000_000)
    xyz, a = synth.dense_cloud(n, seed=1)
    scale = 0.5
    tgt = np.unique(np.rint(xyz.astype(np.float64) * scale).astype(np.int32), axis=0)
    p = recolour_params(bitdepth=8)
    ctx.recolour(p, xyz, a, tgt, scale=scale)  # warm-up (pool)
    ctx.set_profiling(True)
    ctx.kernel_times()
    t0 = time.perf_counter()
    got = ctx.recolour(p, xyz, a, tgt, scale=scale)
    dt = time.perf_counter() - t0
    kt = ctx.kernel_times()
    ctx.set_profiling(False)
    kms = sum(v[0] for v in kt.values())
    nbytes = len(xyz) * (12 + 4 * 3) + len(tgt) * (12 + 4 * 3)
    res = {"workload": f"pcc::recolour, {len(xyz)}-point S-dense colour source -> {len(tgt)} target points (scale 0.5), defaults",
           "call_ms": round(dt * 1e3, 2), "kernel_ms": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
           "value": round(len(tgt) / dt / 1e6, 2), "unit": "M target points/s (host tier, PCIe included)",
           "kernels_GBps": round(nbytes / (kms / 1e3) / 1e9, 2), "bytes": nbytes}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_loader as ol
        chk = ol.ref() if ol.ref_available() else ol.oracle()
        t0 = time.perf_counter()
        ref = chk.recolour(p, xyz, a, tgt, scale=scale)
        res["cpu_baseline"] = {"value": round(len(tgt) / (time.perf_counter() - t0) / 1e6, 3), "unit": "M target points/s",
                               "cores": 1, "kind": "reference" if ol.ref_available() else "port"}
        # the device builds the reference's k-d trees and walks them in its order: equidistant candidates
        # (a dyadic scale on a voxelised cloud has them at nearly every point) come out as the reference has them
        res["identical_to_cpu_fraction"] = round(float(np.all(got == ref, axis=1).mean()), 4)
        res["max_abs_difference"] = int(np.abs(got - ref).max())
    return res


def raht_inter_leg(ctx, args):
    """RAHT with attribute inter prediction (gpcc_raht_forward_inter / _inverse_inter), one
    1 M-point S-lidar reflectance frame predicted from a jittered copy of itself with 10 % of the points gone; the
    reference's default configuration of the tool (sub-node prediction, per-layer inter / intra decision
    rahtEnableCodeLayer = 1, fixed filter taps, skipInitLayersForFiltering = 3).
    Host tier (host buffers in and out, PCIe included).  Algorithmic bytes per point: 8 + 4c in, 4c + 4c out for the
    frame being coded, 8 + 4c in for the reference frame."""
    from mpeg_pcc_tmc13_amd import RahtInterParams, raht_params, synth
    n = min(args.points, 1_000_000)
    rng = np.random.default_rng(1)
    xyz, attrs = synth.lidar_cloud(n, seed=7)
    attrs = (attrs >> 8 if attrs.max() > 255 else attrs)[:, :1]
    morton, a, _ = synth.sort_by_morton(xyz, attrs)
    keep = rng.random(len(xyz)) > 0.1
    xr = np.clip(xyz + rng.integers(-1, 2, size=xyz.shape), 0, None)[keep].astype(np.int32)
    ar = np.clip(attrs + rng.integers(-4, 5, size=attrs.shape), 0, 255)[keep].astype(np.int32)
    mref, aref, _ = synth.sort_by_morton(xr, ar)
    p = raht_params()
    ip = RahtInterParams(15, 1, 0, 3)
    ctx.raht_forward_inter(p, ip, morton, a, mref, aref)  # warm-up (pool, log2 table)
    # wall time without the profiler (with it the encoder's two candidates of a level run one after the other)
    t0 = time.perf_counter()
    co, rec, modes, taps = ctx.raht_forward_inter(p, ip, morton, a, mref, aref)
    t1 = time.perf_counter()
    dec = ctx.raht_inverse_inter(p, ip, morton, co, 1, mref, aref, modes, taps)
    t2 = time.perf_counter()
    ctx.set_profiling(True)
    ctx.kernel_times()
    ctx.raht_forward_inter(p, ip, morton, a, mref, aref)
    kt_f = ctx.kernel_times()
    ctx.raht_inverse_inter(p, ip, morton, co, 1, mref, aref, modes, taps)
    kt_i = ctx.kernel_times()
    ctx.set_profiling(False)

    def agg(kt):
        o = {}
        for name, (ms, _) in kt.items():
            o[name.split("@")[0]] = round(o.get(name.split("@")[0], 0.0) + ms, 3)
        return dict(sorted(o.items(), key=lambda kv: -kv[1])[:8])
    res = {"workload": f"RAHT with attribute inter prediction, {len(morton)}-point S-lidar reflectance frame, reference frame of "
                       f"{len(mref)} points, the reference's default flags (sub-node prediction, per-layer decision, fixed taps), qp 34",
           "forward_ms": round((t1 - t0) * 1e3, 2), "inverse_ms": round((t2 - t1) * 1e3, 2),
           "value": round(2 * len(morton) / (t2 - t0) / 1e6, 2), "unit": "Mpoints/s (forward + inverse, host tier, PCIe included)",
           "layer_modes": modes.tolist(), "filter_taps": taps.tolist(), "decoder_equals_encoder_recon": bool(np.array_equal(dec, rec)),
           "forward_kernels_ms": agg(kt_f), "inverse_kernels_ms": agg(kt_i)}
    if not args.no_cpu_baseline:
        res.update(raht_inter_cpu_baseline(p, morton, a, mref, aref, co, modes, taps))
    return res


def hbm_calibration(torch, dev):
    """What a plain device copy reaches on THIS box (read + write bytes / time,
    1 GiB), next to the nominal peak the roofline fractions are quoted against."""
    nbytes = 1 << 30
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev).fill_(1)
    c = torch.empty_like(a)
    dt = timed(torch, dev, lambda: c.copy_(a), 10, warmup=2)
    return {"copy_GBps": round(2 * nbytes / dt / 1e9, 1), "bytes": nbytes,
            "nominal_peak_GBps": HBM_PEAK_GBPS,
            "note": "torch device-to-device copy, read + write bytes; roofline.frac uses the nominal peak"}


def pmc_traffic(args, kernel, workload=None):
    """HBM-side bytes per launch of a kernel (FETCH_SIZE + WRITE_SIZE of rocprofv3, separate --pmc passes) from the
    committed counter passes of THIS workload at this round's HEAD (profiles/r06_pmc_traffic.json, made by
    tools/pmc.sh + tools/pmc_json.py); None for any other workload or kernel.  PMC passes cannot run inside
    this process.  Raw counter bytes: the guide's x2 correction of FETCH_SIZE applies to wide coalesced streams only,
    and these kernels' traffic is 4..16-byte gathers, polls and write-through granules."""
    path = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
    if workload is None:
        default = (args.cloud == "lidar" and args.points == 1_000_000 and args.frames == 1 and args.subnode == 1
                   and args.qp == 34 and not args.haar and args.direction == "both")
        if not default:
            return None
        workload = "headline"
    if not os.path.exists(path):
        return None
    rec = json.load(open(path)).get(workload, {}).get(kernel)
    if rec is None:
        return None
    return rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]


def lifting_leg(ctx, args, torch=None, dev=None):
    """BASELINE configs[2] on one GPU: the lifting attribute coder of a 5M-point
    dense colour cloud in 5 slices of 1M points minus the entropy loop --
    encoder side = LoD build (kNN predictor search) + lifting forward, decoder
    side = LoD build + lifting inverse -- through the DEVICE tier: positions,
    attributes, coefficients and the LoD structure stay in HBM
    (gpcc_dev_lift_encode_attr / gpcc_dev_lift_decode_attr).  Algorithmic bytes
    (SURVEY.md 8(d)): LoD build 40 B/pt, lifting encode 28 + 12C, decode 28 + 8C."""
    from mpeg_pcc_tmc13_amd import lift_params, lod_params, synth
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    slices = 5 if args.points >= 1_000_000 else 2
    per = min(args.points, 1_000_000)
    clouds = [synth.dense_cloud(per, seed=77 + i, bits=10) for i in range(slices)]
    sizes = [len(c[0]) for c in clouds]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n, c = int(offsets[-1]), 3
    lp = lod_params()
    d_xyz = to_device(torch, np.concatenate([x for x, _ in clouds]), dev)
    src = to_device(torch, np.concatenate([a for _, a in clouds]).reshape(-1), dev)
    d_attrs, d_dec = torch.empty_like(src), torch.empty_like(src)
    d_co = torch.zeros(c * n, dtype=torch.int32, device=dev)
    d_lod = [torch.zeros(k * n, dtype=torch.int32, device=dev) for k in (1, 3, 3, 1)]
    ctx.set_morton_bits(30)

    def encode():
        d_attrs.copy_(src)
        torch.cuda.synchronize(dev)
        lfs = [lift_params([s], qp=34) for s in sizes]
        t0 = time.perf_counter()
        lcp = ctx.dev_lift_attr(True, lp, lfs, offsets, d_xyz.data_ptr(), d_attrs.data_ptr(), d_co.data_ptr(), c)
        return time.perf_counter() - t0, lcp, lfs

    def decode(lcp):
        lfs = [lift_params([s], qp=34) for s in sizes]
        t0 = time.perf_counter()
        ctx.dev_lift_attr(False, lp, lfs, offsets, d_xyz.data_ptr(), d_dec.data_ptr(), d_co.data_ptr(), c, lcp=lcp)
        return time.perf_counter() - t0

    def lod_only():
        t0 = time.perf_counter()
        ctx.dev_lod_build(lp, offsets, d_xyz.data_ptr(), *[t.data_ptr() for t in d_lod])
        return time.perf_counter() - t0

    encode()  # warm-up: arena, code objects
    t_enc, lcp, lfs = min((encode() for _ in range(2)), key=lambda r: r[0])
    t_dec = min(decode(lcp) for _ in range(2))
    t_lod = min(lod_only() for _ in range(2))
    ok = bool(torch.equal(d_attrs, d_dec))
    ctx.set_profiling(True)
    ctx.kernel_times()
    lod_only()
    kt = ctx.kernel_times()
    ctx.set_profiling(False)
    name, (ms, launches) = max(kt.items(), key=lambda kv: kv[1][0])
    res = {"workload": f"{slices} x {per}-point S-dense colour slices resident in HBM, {lfs[0].num_lods} LoDs, "
                       "distance sub-sampling, 3 neighbours, qp 34 (device tier)",
           "encode_ms": round(t_enc * 1e3, 2), "decode_ms": round(t_dec * 1e3, 2),
           "lod_build_alone_ms": round(t_lod * 1e3, 2),
           "lod_build_ms_per_Mpoint": round(t_lod * 1e3 / (n / 1e6), 2),
           "value": round(n / (t_enc + t_dec) / 1e6, 3), "unit": "Mpoints/s (encode + decode)",
           "roundtrip_decoder_equals_encoder_recon": ok,
           "roofline": {"bound": "hbm", "kernel": name,
                        "achieved": round(40 * n / (ms / 1e3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(40 * n / (ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 5),
                        "algorithmic_bytes_per_point": {"lod_build": 40, "lift_encode": 28 + 12 * c, "lift_decode": 28 + 8 * c},
                        "pipeline_achieved": {"lod_build": round(40 * n / t_lod / 1e9, 2),
                                              "encode": round((40 + 28 + 12 * c) * n / t_enc / 1e9, 2),
                                              "decode": round((40 + 28 + 8 * c) * n / t_dec / 1e9, 2)},
                        "lod_kernel_ms": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:8]}}}
    ctx.set_morton_bits(0)
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import lod_helpers as lh
        import oracle_loader as ol
        kind = "reference" if ol.ref_available() else "port"
        xyz0 = clouds[0][0]
        t0 = time.perf_counter()
        o = (lh.ref_lod_generate if kind == "reference" else lh.oracle_lod_generate)(xyz0, lp)
        t_ref = time.perf_counter() - t0
        g = ctx.lod_build(lp, xyz0)
        same = all(np.array_equal(np.asarray(g[k]).astype(np.int64), np.asarray(o[k]).astype(np.int64))
                   for k in ("npl", "indexes", "nc", "ni", "w"))
        res["cpu_lod_build"] = {"value": round(len(xyz0) / t_ref / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": kind,
                                "sample": f"AttributeLods::generate on slice 0 ({len(xyz0)} points), {t_ref:.2f} s "
                                          "(needed once by the encoder and once by the decoder)",
                                "gpu_result_identical": bool(same)}
    return res


def predicting_leg(ctx, args):
    """The other half of BASELINE configs[2]: the PREDICTING transform of a
    1M-point dense colour slice, CTC tools (three direct predictors, inter-
    component prediction, intra-LoD prediction, quantNeighWeight 16/8/4) --
    the decoder (gpcc_pred_inverse) and the encoder (gpcc_pred_forward: the
    choice among the direct predictors iterated to the sequential coder's fixed
    point, `encode_passes` DAG passes).  Host tier: the timings are the kernels'
    (HIP events), the CPU figure is the same loop of the oracle (pinned to the
    reference at symbol level) on one core."""
    from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth
    n = min(args.points, 1_000_000)
    xyz, attrs = synth.dense_cloud(n, seed=41, bits=10 if n >= 500_000 else 8)
    n = len(xyz)
    lp = lod_params(levels=12, lifting=False, intra_range=1100000, blend=True)
    lp.intra_lod_prediction_skip_layers = 0
    lod = ctx.lod_build(lp, xyz)
    qnw = (16, 8, 4)
    pp = pred_params(lod["npl"], qp=28, bitdepth=8, max_levels=12, quant_neigh_weight=qnw)
    pp0 = pp  # (until round 3 the device encoder declined direct predictors and ran without)
    res = {"workload": f"predicting transform, {n}-point S-dense colour slice, {len(lod['npl'])} LoDs, intra-LoD "
                       "prediction, 3 direct predictors, ICP, qp 28 (host tier, kernel times)"}
    values = icp = want = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import lod_helpers as lh
        t0 = time.perf_counter()
        values, want, icp, modes = lh.oracle_pred(True, pp, lod, attrs=attrs)
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        lh.oracle_pred(False, pp, lod, values=values, icp=icp)
        t_dec = time.perf_counter() - t0
        res["cpu_port"] = {"encode_ms": round(t_enc * 1e3, 1), "decode_ms": round(t_dec * 1e3, 1), "cores": 1,
                           "kind": "port", "direct_modes_chosen": int((modes > 0).sum())}
        import oracle_loader as ol
        if ol.ref_available():
            # the compiled reference's whole operator for the same slice: AttributeLods::generate (twice),
            # encodeColorsPred + decodeColorsPred and its arithmetic coder
            t0 = time.perf_counter()
            payload, rec_enc, rec_dec, _ = lh.ref_pred_roundtrip(lp, pp, 64, 28, 0, xyz, attrs)
            t_op = time.perf_counter() - t0
            res["cpu_reference_operator"] = {
                "seconds": round(t_op, 3), "cores": 1, "kind": "reference", "payload_bytes": len(payload),
                "what": "AttributeEncoder::encode + AttributeDecoder::decode (LoD generation x2, transform, entropy coder)",
                "reconstruction_equals_port": bool(np.array_equal(rec_enc, want) and np.array_equal(rec_dec, want))}
    # encoder without direct predictors, then the decoder on its symbols
    ctx.pred_forward(pp0, lod["nc"], lod["ni"], lod["w"], lod["indexes"], attrs)  # warm-up
    ctx.set_profiling(True)
    ctx.kernel_times()
    v0, rec0, icp0 = ctx.pred_forward(pp0, lod["nc"], lod["ni"], lod["w"], lod["indexes"], attrs)
    kte_raw = ctx.kernel_times()
    kt_e = {k: v[0] for k, v in kte_raw.items() if k.startswith("pred")}
    res["encode_passes"] = int(round(kte_raw.get("pred_dag", (0, 0))[1]))
    dec0 = ctx.pred_inverse(pp0, lod["nc"], lod["ni"], lod["w"], lod["indexes"], v0, icp=icp0)
    kt_d = {k: v[0] for k, v in ctx.kernel_times().items() if k.startswith("pred")}
    ok = bool(np.array_equal(dec0, rec0))
    if values is not None:
        # the CTC stream (direct predictors chosen by the reference's serial mode decision): decoder only
        got = ctx.pred_inverse(pp, lod["nc"], lod["ni"], lod["w"], lod["indexes"], values, icp=icp)
        kt_d = {k: v[0] for k, v in ctx.kernel_times().items() if k.startswith("pred")}
        ok = ok and bool(np.array_equal(got, want)) and bool(np.array_equal(v0, values)) and bool(np.array_equal(rec0, want))
    ctx.set_profiling(False)
    c = 3
    res.update({"encode_kernels_ms": {k: round(v, 3) for k, v in kt_e.items()},
                "decode_kernels_ms": {k: round(v, 3) for k, v in kt_d.items()},
                "decode_value": round(n / (sum(kt_d.values()) / 1e3) / 1e6, 1), "unit": "Mpoints/s (kernels)",
                "encode_value": round(n / (sum(kt_e.values()) / 1e3) / 1e6, 1),
                "algorithmic_bytes_per_point": {"decode": 28 + 8 * c, "encode": 28 + 12 * c},
                "decode_achieved_GBps": round((28 + 8 * c) * n / (sum(kt_d.values()) / 1e3) / 1e9, 2),
                "results_identical": ok})
    # configs[2] shape through the device tier: 5 slices resident in HBM, LoD build + transform per
    # slice, slices concurrent on the context's lanes (wall clock, LoD build included)
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    slices = 5 if args.points >= 1_000_000 else 2
    clouds = [synth.dense_cloud(n, seed=41 + i, bits=10 if n >= 500_000 else 8) for i in range(slices)]
    sizes = [len(cl[0]) for cl in clouds]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    tot = int(offs[-1])
    d_xyz = to_device(torch, np.concatenate([cl[0] for cl in clouds]), dev)
    src = to_device(torch, np.concatenate([cl[1] for cl in clouds]).reshape(-1), dev)
    d_attrs = torch.empty_like(src)
    d_vals = torch.zeros(3 * tot, dtype=torch.int32, device=dev)
    d_dec = torch.zeros(3 * tot, dtype=torch.int32, device=dev)
    ctx.set_morton_bits(30)
    mk = lambda: [pred_params([sz], qp=28, bitdepth=8, max_levels=12, quant_neigh_weight=qnw) for sz in sizes]

    def enc():
        d_attrs.copy_(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        l = ctx.dev_pred_attr(True, lp, mk(), offs, d_xyz.data_ptr(), d_attrs.data_ptr(), d_vals.data_ptr(), 3)
        return time.perf_counter() - t0, l

    def dec(l):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.dev_pred_attr(False, lp, mk(), offs, d_xyz.data_ptr(), d_dec.data_ptr(), d_vals.data_ptr(), 3, icp=l)
        return time.perf_counter() - t0
    enc()
    t_e, l = min((enc() for _ in range(2)), key=lambda r: r[0])
    t_d = min(dec(l) for _ in range(2))
    ctx.set_morton_bits(0)
    res["device_tier"] = {"workload": f"{slices} x {sizes[0]}-point slices resident in HBM, LoD build + transform per slice",
                          "encode_ms": round(t_e * 1e3, 2), "decode_ms": round(t_d * 1e3, 2),
                          "value": round(tot / (t_e + t_d) / 1e6, 3), "unit": "Mpoints/s (encode + decode, LoD build included)",
                          "roundtrip_decoder_equals_encoder_recon": bool(torch.equal(d_attrs, d_dec))}
    # the encoder's fixed-point iteration over everything this leg coded: a slice that does not settle
    # within the pass limit (64) would be declined to the CPU
    res["encoder_pass_statistics"] = ctx.pred_pass_stats()
    return res


_CPU = {}


def _cpu_init(path):
    """Start-up of one process of the all-cores CPU baseline (outside the
    timing): load the checker library and the frame."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    ge.load_package()
    import oracle_loader as ol
    from mpeg_pcc_tmc13_amd import raht_params
    z = np.load(path)
    _CPU["p"] = raht_params(qp=int(z["qp"]), subnode=bool(z["subnode"]), search_range=int(z["search_range"]))
    _CPU["chk"] = ol.ref() if ol.ref_available() else ol.oracle()
    _CPU["morton"], _CPU["attrs"] = np.ascontiguousarray(z["morton"]), np.ascontiguousarray(z["attrs"])


def _cpu_worker(_):
    """forward + inverse of the frame with the compiled reference (or the C port)"""
    t0 = time.perf_counter()
    co, rec = _CPU["chk"].raht_forward(_CPU["p"], _CPU["morton"], _CPU["attrs"])
    _CPU["chk"].raht_inverse(_CPU["p"], _CPU["morton"], co, _CPU["attrs"].shape[1])
    return time.perf_counter() - t0


def raht_inter_cpu_baseline(p, morton, a, mref, aref, co, modes, taps):
    """cpu_baseline of raht_inter_leg: the compiled reference (or the oracle) on the same frame, forward only;
    the checker is used for nothing else"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_loader as ol
    from test_oracle_raht_inter import run
    lib, fn, kind = (ol.ref().lib, "ref_raht_inter", "reference") if ol.ref_available() else (ol.oracle().lib, "oracle_raht_inter", "port")
    t0 = time.perf_counter()
    rc, co_r, _, modes_r, taps_r = run(lib, fn, p, True, morton, a, None, mref, aref, 15, 1, 0, 3)
    dt = time.perf_counter() - t0
    return {"cpu_baseline": {"value": round(len(morton) / dt / 1e6, 3), "unit": "Mpoints/s (forward)", "cores": 1, "kind": kind,
                             "sample": f"the same frame, forward only, {dt:.2f} s"},
            "identical_to_cpu": bool(rc == 0 and np.array_equal(co, co_r) and np.array_equal(modes, modes_r) and np.array_equal(taps, taps_r))}


def cpu_baseline(frame, p, c):
    """oracle/_ref (the reference's own RAHT.cpp, -O3) forward+inverse on the
    same frame: one core, then every host core (one process per frame -- the
    reference is serial code, slices / frames are its unit of parallelism);
    falls back to the C oracle port if the compiled reference did not travel."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_loader as ol
    morton, attrs, _ = frame
    kind = "reference" if ol.ref_available() else "port"
    chk = ol.ref() if kind == "reference" else ol.oracle()
    n = len(morton)
    # bounded sample: the whole frame, repeated until about 8 s of CPU work
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while reps < 12 and dt < 8.0:
        co, rec = chk.raht_forward(p, morton, attrs)
        chk.raht_inverse(p, morton, co, c)
        reps += 1
        dt = time.perf_counter() - t0
    res = {"value": round(n * reps / dt / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": kind,
           "sample": f"forward+inverse of frame 0 ({n} points, same flags) x {reps}, "
                     f"{dt:.1f} s wall, host {os.cpu_count()} logical cores, 1 used"}
    # all cores: one process per core, each transforming the same frame
    try:
        import multiprocessing as mp
        import tempfile
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        logical = cores
        # one process per core up to 64: the reference is memory-bound when every
        # logical core runs a frame (measured on this box: 256 processes take 46 s
        # each against 1.6 s alone, 5.1 Mpoints/s in total) and the bench has to
        # finish within minutes; GPCC_BENCH_CPU_PROCS overrides
        cores = min(cores, int(os.environ.get("GPCC_BENCH_CPU_PROCS", "64")))
        try:  # the reference keeps ~0.5 GB per 1M-point frame: stay far inside the host's memory
            import psutil
            cores = max(1, min(cores, int(psutil.virtual_memory().available // (2 << 30))))
        except ImportError:
            cores = min(cores, 32)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "frame.npz")
            np.savez(path, morton=morton, attrs=attrs, qp=p.layer_qp[0][0],
                     subnode=p.raht_subnode_prediction_enabled_flag,
                     search_range=p.raht_prediction_search_range)
            mpc = mp.get_context("spawn")  # a fresh interpreter per worker: no HIP state is inherited
            with mpc.Pool(cores, initializer=_cpu_init, initargs=(path,)) as pool:
                t0 = time.perf_counter()
                busy = pool.map(_cpu_worker, range(cores), chunksize=1)
                wall = time.perf_counter() - t0
        res["all_cores"] = {"value": round(n * cores / wall / 1e6, 3), "unit": "Mpoints/s", "cores": cores,
                            "logical_cores": logical,
                            "sample": f"{cores} processes x 1 forward+inverse of the same frame, {wall:.1f} s wall "
                                      f"(mean {np.mean(busy):.1f} s per process)"}
    except Exception as e:  # the one-core figure stands on its own
        res["all_cores"] = {"error": repr(e)}
    return res


if __name__ == "__main__":
    main()

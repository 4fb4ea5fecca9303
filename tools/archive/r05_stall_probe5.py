#!/usr/bin/env python3
"""Fifth probe: is the slow step the library's, or the box's?  Three loops of 12 s each, every one with a second
thread that only reads the clock (a gap in ITS readings = that thread did not run: the host side froze):
  A  no GPU work at all (the main thread sleeps 1 ms per step)
  B  a trivial torch kernel + torch.cuda.synchronize per step (nothing of this library)
  C  the library's sub-node-off forward over 5 slices per step (the leg with the slow steps)
For every loop: steps, median, the steps above 3x median + 3 ms, the clock thread's gaps above 2 ms, and how many of
the slow steps overlap a gap."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    so = "/tmp/r05_stall_stack.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", os.path.join(ROOT, "tools", "r05_stall_stack.c"),
                           "-o", so, "-lpthread"])
    dog = C.CDLL(so)
    dog.spin_gaps.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int]
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    from mpeg_pcc_tmc13_amd import context, raht_params
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(0, stream=stream.cuda_stream)
    frames = [bench.make_frame("lidar", 1_000_000, seed=1 + i) for i in range(5)]
    p = raht_params(qp=34, subnode=False, search_range=2500)
    b = bench.Batch(torch, dev, ctx, frames, p)
    x = torch.zeros(1 << 20, device=dev)
    secs = float(os.environ.get("PROBE_SECONDS", "12"))

    def loop(step):
        cap = 4096
        buf = (C.c_double * (2 * cap))()
        got = {}
        th = threading.Thread(target=lambda: got.setdefault("n", dog.spin_gaps(secs + 0.5, 2.0, buf, cap)))
        th.start()
        time.sleep(0.2)
        ts = []
        t_end = time.monotonic() + secs
        while time.monotonic() < t_end:
            t0 = time.monotonic()
            step()
            ts.append((t0, time.monotonic() - t0))
        th.join()
        gaps = [(buf[2 * i], buf[2 * i + 1]) for i in range(got["n"])]
        med = sorted(d for _, d in ts)[len(ts) // 2]
        slow = [(t0, d) for t0, d in ts if d > 3 * med + 0.003]
        hit = sum(1 for t0, d in slow if any(g0 < t0 + d and g0 + gl / 1e3 > t0 for g0, gl in gaps))
        return {"steps": len(ts), "median_ms": round(med * 1e3, 3), "slow_steps": len(slow),
                "slow_ms": [round(d * 1e3, 1) for _, d in slow][:24],
                "clock_thread_gaps": len(gaps), "gaps_ms": [round(g, 1) for _, g in gaps][:24],
                "slow_steps_overlapping_a_gap": hit}

    def step_a():
        time.sleep(0.001)

    def step_b():
        x.add_(1.0)
        torch.cuda.synchronize(dev)

    def step_c():
        b.forward()
        torch.cuda.synchronize(dev)

    for _ in range(3):
        step_b(); step_c()
    res = {"cpus": os.cpu_count(), "A_no_gpu": loop(step_a), "B_torch_only": loop(step_b), "C_library_sub0_forward_5": loop(step_c),
           "B_again": loop(step_b)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

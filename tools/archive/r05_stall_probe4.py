#!/usr/bin/env python3
"""Fourth probe: WHERE is the host thread during the slow step?  The sequence of probe 3 without the per-kernel timers;
a watchdog thread (tools/r05_stall_stack.c, compiled here) signals the main thread once a sub-node-off forward step has
lasted 8 ms (usual: 0.4-3.5 ms) and the handler writes the native stack.  Also: how long a 100 us nanosleep takes here."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out_dir = os.path.join(ROOT, "gpurun_out", "r05_stall_stack")
    os.makedirs(out_dir, exist_ok=True)
    so = "/tmp/r05_stall_stack.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", os.path.join(ROOT, "tools", "r05_stall_stack.c"),
                           "-o", so, "-lpthread"])
    dog = C.CDLL(so)
    dog.sleep_probe_us.restype = C.c_double
    dog.dog_start.argtypes = [C.c_char_p, C.c_double, C.c_double]
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    from mpeg_pcc_tmc13_amd import context, raht_params
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = context(0, stream=stream.cuda_stream)
    frames = [bench.make_frame("lidar", 1_000_000, seed=1 + i) for i in range(10)]
    res = {"nanosleep_100us_takes_us": round(dog.sleep_probe_us(200), 1), "slow": []}
    dog.dog_start(os.path.join(out_dir, "stacks.txt").encode(), 8.0, 4.0)
    for rep in range(int(os.environ.get("PROBE_REPS", "8"))):
        for sub in (1, 0):
            p = raht_params(qp=34, subnode=bool(sub), search_range=2500)
            for nf in (1, 2, 5, 10):
                b = bench.Batch(torch, dev, ctx, frames[:nf], p)
                for fn_name in ("forward", "inverse"):
                    fn = getattr(b, fn_name)
                    fn(); fn()
                    torch.cuda.synchronize(dev)
                    if float(os.environ.get("PROBE_SETTLE", "0")) > 0:  # (let whatever the allocations set off finish)
                        time.sleep(float(os.environ["PROBE_SETTLE"]))
                    ts = []
                    for i in range(10):
                        watch = sub == 0 and fn_name == "forward"
                        t0 = time.perf_counter()
                        if watch:
                            dog.dog_step_begin()
                        fn()
                        t1 = time.perf_counter()
                        torch.cuda.synchronize(dev)
                        if watch:
                            dog.dog_step_end()
                        t2 = time.perf_counter()
                        ts.append((t2 - t0, t1 - t0))
                    med = sorted(t for t, _ in ts)[5]
                    for i, (dt, dcall) in enumerate(ts):
                        if dt > 3 * med + 0.003:
                            res["slow"].append({"rep": rep, "subnode": sub, "slices": nf, "direction": fn_name, "step": i,
                                                "wall_ms": round(dt * 1e3, 2), "in_call_ms": round(dcall * 1e3, 2),
                                                "median_ms": round(med * 1e3, 3)})
                del b
    dog.dog_stop()
    # the modules' load addresses, to resolve the offsets offline
    with open("/proc/self/maps") as f:
        maps = [l for l in f if " r-xp " in l and (".so" in l)]
    with open(os.path.join(out_dir, "maps.txt"), "w") as f:
        f.writelines(maps)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

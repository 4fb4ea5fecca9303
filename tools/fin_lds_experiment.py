#!/usr/bin/env python3
"""The finish kernel's tables in LDS (the form that gave wrong duplicate-chain coefficients on the
MI355X in round 3) against the shipped form, as experiment builds of the library:

    python tools/fin_lds_experiment.py build          # here: exp/libgpcc_fin<V>.so, V in 0 1 2 4 5
    python tools/fin_lds_experiment.py run [reps]     # on the GPU box: every build in a child process
    python tools/fin_lds_experiment.py child <V> <reps>

GPCC_FIN_VAR (6: as 1 with the source attributes read past the L1, 7: as 1 with every store drained): 0 tables in global memory (shipped), 1 staged in LDS, 2 the same + checks of the
staged copy against the global tables behind the barrier and when the threads leave, 4 the (w, 1)
butterfly's constant operand opaque (nothing hoisted to the scalar unit), 5 the kernel unoptimised, 8 as 1 in a 64-register allocation (clobber of v63).
Each child runs the six batches pinned in tests/test_gpu_batches.py `reps` times against the compiled
reference and prints one JSON line."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARS = [int(x) for x in os.environ.get('FIN_VARS', '0 1 2 4 5').split()]


def lib(v):
    return os.path.join(ROOT, "exp", "libgpcc_fin%d.so" % v)


def build():
    os.makedirs(os.path.join(ROOT, "exp"), exist_ok=True)
    procs = []
    for v in VARS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mpeg-pcc-tmc13_amd", "csrc"),
               "-DGPCC_FIN_VAR=%d" % v, os.path.join(ROOT, "mpeg-pcc-tmc13_amd", "csrc", "gpcc_attr_mi355.hip"),
               "-o", lib(v)]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0


def detail(ctx, p, c, ms, as_, want, i):
    """one more run of the batch; the differing coefficients of slice i with the place of each in
    its leaf's chain of duplicates (leaf, weight, steps from the chain's start)"""
    import numpy as np, torch
    sizes = [len(m) for m in ms]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    dev = torch.device("cuda:0")
    d_m = torch.from_numpy(np.concatenate(ms)).to(dev)
    d_a = torch.from_numpy(np.concatenate(as_).reshape(-1)).to(dev)
    d_c = torch.zeros(c * int(offsets[-1]), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.dev_raht_forward(p, offsets, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), c)
    ctx.synchronize()
    co = d_c.cpu().numpy(); rec = d_a.cpu().numpy()
    b, n = int(offsets[i]), sizes[i]
    got, ref = co[c * b:c * (b + n)], want[i][0]
    m = ms[i]
    uniq, first, cnt = np.unique(m, return_index=True, return_counts=True)
    nu = len(uniq)
    # chain of leaf l: coefficients nu + first[l] - l .. + cnt[l] - 2  (weights cnt-1 .. 1)
    starts = nu + first - np.arange(nu)
    d = np.nonzero(got[:n] != ref[:n])[0]
    rows = []
    for x in d[:24]:
        l = int(np.searchsorted(starts, x, side="right") - 1) if x >= nu else -1
        rows.append([int(x), l, int(cnt[l]) if l >= 0 else 0, int(x - starts[l]) if l >= 0 else -1, int(got[x]), int(ref[x])])
    gr, rr = rec[c * b:c * (b + n)], want[i][1].reshape(-1)
    dr = np.nonzero(gr != rr)[0]
    return {"slice": i, "n": n, "unique": int(nu), "coeff_diffs": int(len(d)), "rows[idx,leaf,weight,step,got,want]": rows,
            "recon_diffs": int(len(dr)), "recon_first": [[int(x), int(gr[x]), int(rr[x])] for x in dr[:12]],
            "src_first": [int(v) for v in as_[i].reshape(-1)[dr[:12]]]}


def child(v, reps):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "stress"))
    import ctypes as C
    import __graft_entry__ as g; g.load_package()
    import oracle_loader as ol
    import stress_cx_batch as sb
    from mpeg_pcc_tmc13_amd import context, _lib
    ctx = context(0); o = ol.ref()
    out = {"var": v, "reps": reps, "runs": 0, "bad_runs": 0, "first": []}
    for it in (46, 144, 259, 404, 449, 483):
        p, c, ms, as_ = sb.make_batch(424200, it)
        want = [o.raht_forward(p, ms[i], as_[i]) for i in range(len(ms))]
        for rep in range(reps):
            bad = sb.run_batch(ctx, o, p, c, ms, as_, want)
            out["runs"] += 1
            if bad:
                out["bad_runs"] += 1
                if len(out["first"]) < 6:
                    out["first"].append([it, rep] + [list(b) for b in bad[:3]])
                if len(out.setdefault("detail", [])) < 4:
                    out["detail"].append(detail(ctx, p, c, ms, as_, want, bad[0][1]))
    if v == 2:
        buf = (C.c_uint32 * (4 + 4 * 60))()
        assert _lib.load().gpcc_debug_fin(buf, 0) == 0
        out["checks"], out["bad_after_barrier"], out["bad_at_end"] = buf[0], buf[1], buf[2]
        out["records"] = [[buf[4 + 4 * k] >> 16, buf[4 + 4 * k] & 0xffff, buf[5 + 4 * k], hex(buf[6 + 4 * k]), hex(buf[7 + 4 * k])]
                          for k in range(min(60, buf[1] + buf[2]))]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        reps = sys.argv[2] if len(sys.argv) > 2 else "6"
        for v in VARS:
            env = dict(os.environ, GPCC_LIB_PATH=lib(v))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(v), reps], env=env,
                               capture_output=True, text=True, timeout=900)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "var %d: no output: %s" % (v, r.stderr[-400:]), flush=True)

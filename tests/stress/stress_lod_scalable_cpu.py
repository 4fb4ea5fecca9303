"""Randomised differential stress of the scalable-lifting LoD build on the CPU (not collected by pytest):
the oracle against the compiled reference, and the library's level loop + kernels under the wavefront
emulator (tests/emu) against the oracle.
    python tests/stress/stress_lod_scalable_cpu.py <seed base> [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g; g.load_package()
import numpy as np
import emu_lod_loader as el, lod_helpers as lh, oracle_loader as ol
from mpeg_pcc_tmc13_amd import lod_params, synth
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
t0 = time.time(); cases = 0
for seed in range(100000):
    rng = np.random.default_rng(int(sys.argv[1]) + seed)
    n = int(rng.integers(1, 40000)) if seed % 5 == 0 else int(rng.integers(1, 3000))
    kind = rng.integers(3)
    if kind == 0: xyz, _ = synth.random_cloud(n, seed=int(rng.integers(1 << 30)), bits=int(rng.integers(2, 21)), dup_fraction=float(rng.choice([0.0, 0.2])))
    elif kind == 1: xyz, _ = synth.dense_cloud(n, seed=int(rng.integers(1 << 30)), bits=int(rng.integers(6, 11)))
    else: xyz, _ = synth.lidar_cloud(n, seed=int(rng.integers(1 << 30)))
    lifting = bool(rng.integers(3) > 0)
    kw = dict(neighbours=int(rng.integers(1, 4)), lifting=lifting, distribution=bool(rng.integers(2)),
              bias=tuple(int(x) for x in rng.integers(1, 4, size=3)) if rng.integers(2) else (1, 1, 1),
              inter_range=int(rng.choice([4, 64, 128, 1100000])), intra_range=int(rng.choice([0, 8, 64])),
              blend=(not lifting) and bool(rng.integers(2)))
    lp = lod_params(**kw)
    lp.intra_lod_prediction_skip_layers = int(rng.choice([0, 2, 5, 0x7fffffff]))
    lp.scalable_lifting_enabled_flag = 1
    lp.max_neigh_range_minus1 = int(rng.choice([0, 1, 5, 50, 5000]))
    o = lh.oracle_lod_generate(xyz, lp)
    if ol.ref_available():
        r = lh.ref_lod_generate(xyz, lp)
        for k in r:
            assert np.array_equal(o[k], r[k]), ("oracle vs reference", k, seed, kw, n)
    el.assert_same_lod(el.scalable_build(lp, xyz), o, f"emulator vs oracle seed {seed} {kw} n={n}")
    cases += 1
    if time.time() - t0 > budget: break
print("scalable lod stress ok", cases, "cases", round(time.time() - t0, 1), "s")

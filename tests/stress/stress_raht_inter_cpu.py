"""Randomised differential stress of RAHT with attribute inter prediction on the CPU (not collected by
pytest): the oracle against the compiled reference -- coefficients, reconstruction, decoder, layer modes,
filter taps.      python tests/stress/stress_raht_inter_cpu.py <seed base> [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g; g.load_package()
import numpy as np
import test_oracle_raht_inter as t
from mpeg_pcc_tmc13_amd import raht_params, synth
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
t0 = time.time(); cases = 0
for seed in range(100000):
    rng = np.random.default_rng(int(sys.argv[1]) + seed)
    n = int(rng.integers(1, 20000)) if seed % 5 == 0 else int(rng.integers(1, 2500))
    kind = rng.integers(3)
    if kind == 0: xyz, attrs = synth.random_cloud(n, seed=int(rng.integers(1 << 30)), bits=int(rng.integers(2, 16)), dup_fraction=float(rng.choice([0.0, 0.3])))
    elif kind == 1: xyz, attrs = synth.dense_cloud(n, seed=int(rng.integers(1 << 30)), bits=int(rng.integers(5, 11)))
    else: xyz, attrs = synth.lidar_cloud(n, seed=int(rng.integers(1 << 30)))
    if attrs.max() > 255: attrs = attrs >> 8
    morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
    if rng.integers(5) == 0:   # an unrelated frame
        fx, fa = synth.random_cloud(int(rng.integers(1, 3000)), seed=int(rng.integers(1 << 30)), bits=int(rng.integers(2, 12)))
        if fa.shape[1] != attrs.shape[1]: fa = np.repeat(fa[:, :1], attrs.shape[1], axis=1)
        mref, aref = synth.sort_by_morton(fx, np.clip(fa, 0, 255).astype(np.int32))[:2]
    else:
        mref, aref = t.frame_of(xyz, attrs, rng, amp=int(rng.choice([0, 1, 3])), drop=float(rng.choice([0.0, 0.1, 0.7])),
                                jitter=int(rng.choice([0, 2, 10, 60])), shift=int(rng.choice([0, 0, 1, 40, 5000])))
    haar = bool(rng.integers(5) == 0)
    kw = dict(haar=haar, qp=4 if haar else int(rng.integers(4, 52)), chroma_offset=int(rng.integers(-3, 2)), prediction=bool(rng.integers(4) > 0), subnode=bool(rng.integers(2)),
              extension=bool(rng.integers(4) > 0), search_range=int(rng.choice([8, 2500, 50000])), threshold0=int(rng.integers(0, 4)),
              threshold1=int(rng.integers(0, 8)))
    depth = int(rng.choice([0, 1, 3, 7, 15])); rdo = int(rng.integers(2)); fest = int(rng.integers(2)); skip = int(rng.choice([0, 1, 3]))
    t.check(raht_params(**kw), morton, a_sorted, mref, aref, depth, rdo, fest, skip, f"seed {seed} n={n} {kw} depth{depth} rdo{rdo} fest{fest} skip{skip}")
    cases += 1
    if time.time() - t0 > budget: break
print("inter raht stress ok", cases, "cases", round(time.time() - t0, 1), "s")

"""Randomised differential stress of RAHT with attribute inter prediction on the MI355X (not collected by pytest):
gpcc_raht_forward_inter / _inverse_inter against the oracle -- coefficients, reconstruction, decoder, layer modes,
filter taps -- over random clouds, frames, kernels and tool settings (the generator of stress_raht_inter_cpu.py).
A declined call (GPCC_ERR_UNSUPPORTED) must be one of the documented cases.
      python tests/stress/stress_raht_inter_gpu.py <seed base> [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g; g.load_package()
import numpy as np
import oracle_loader as ol
import test_oracle_raht_inter as t
from mpeg_pcc_tmc13_amd import RahtInterParams, context, raht_params, synth
from mpeg_pcc_tmc13_amd._lib import GpccError
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
ctx = context(0)
t0 = time.time(); cases = declined = 0
for seed in range(100000):
    rng = np.random.default_rng(int(sys.argv[1]) + seed)
    n = int(rng.integers(2, 60000)) if seed % 5 == 0 else int(rng.integers(2, 2500))
    kind = rng.integers(3)
    if kind == 0: xyz, attrs = synth.random_cloud(n, seed=int(rng.integers(1 << 30)), bits=int(rng.integers(2, 16)), dup_fraction=float(rng.choice([0.0, 0.3])))
    elif kind == 1: xyz, attrs = synth.dense_cloud(n, seed=int(rng.integers(1 << 30)), bits=int(rng.integers(5, 11)))
    else: xyz, attrs = synth.lidar_cloud(n, seed=int(rng.integers(1 << 30)))
    if attrs.max() > 255: attrs = attrs >> 8
    morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
    if len(morton) < 2: continue
    if rng.integers(5) == 0:   # an unrelated frame
        fx, fa = synth.random_cloud(int(rng.integers(1, 3000)), seed=int(rng.integers(1 << 30)), bits=int(rng.integers(2, 12)))
        if fa.shape[1] != attrs.shape[1]: fa = np.repeat(fa[:, :1], attrs.shape[1], axis=1)
        mref, aref = synth.sort_by_morton(fx, np.clip(fa, 0, 255).astype(np.int32))[:2]
    else:
        mref, aref = t.frame_of(xyz, attrs, rng, amp=int(rng.choice([0, 1, 3])), drop=float(rng.choice([0.0, 0.1, 0.7])),
                                jitter=int(rng.choice([0, 2, 10, 60])), shift=int(rng.choice([0, 0, 1, 40, 5000])))
    haar = bool(rng.integers(4) == 0)
    kw = dict(haar=haar, qp=4 if haar else int(rng.integers(4, 52)), chroma_offset=int(rng.integers(-3, 2)), prediction=bool(rng.integers(4) > 0), subnode=bool(rng.integers(2)),
              extension=bool(rng.integers(4) > 0), search_range=int(rng.choice([8, 2500, 50000])), threshold0=int(rng.integers(0, 4)),
              threshold1=int(rng.integers(0, 8)))
    depth = int(rng.choice([0, 1, 3, 7, 15])); rdo = int(rng.integers(2)); fest = int(rng.integers(2)); skip = int(rng.choice([0, 1, 3]))
    tag = f"seed {seed} n={n} {kw} depth{depth} rdo{rdo} fest{fest} skip{skip}"
    p = raht_params(**kw); ip = RahtInterParams(depth, rdo, fest, skip)
    rc, co_o, rec_o, modes_o, taps_o = t.run(ol.oracle().lib, "oracle_raht_inter", p, True, morton, a_sorted, None, mref, aref, depth, rdo, fest, skip)
    assert rc == 0, tag
    try:
        co, rec, modes, taps = ctx.raht_forward_inter(p, ip, morton, a_sorted, mref, aref)
    except GpccError as e:
        delta = (int(mref[0] ^ mref[-1]).bit_length() - int(morton[0] ^ morton[-1]).bit_length()) if len(mref) > 1 else 0
        assert e.code == -2 and haar and delta % 3, (tag, e)
        declined += 1
        continue
    assert np.array_equal(taps, taps_o), (tag, "taps", taps, taps_o)
    assert np.array_equal(modes, modes_o), (tag, "modes", modes, modes_o)
    assert np.array_equal(co, co_o), (tag, "coefficients")
    assert np.array_equal(rec, rec_o), (tag, "reconstruction")
    dec = ctx.raht_inverse_inter(p, ip, morton, co_o, a_sorted.shape[1], mref, aref, modes_o, taps_o)
    assert np.array_equal(dec, rec_o), (tag, "decoder")
    cases += 1
    if time.time() - t0 > budget: break
print("inter raht gpu stress ok", cases, "cases,", declined, "declined (Haar, trees not aligned)", round(time.time() - t0, 1), "s")

"""Seeded cases of pcc::recolour shared by tests/golden/make_recolour_golden.py (which records the COMPILED
REFERENCE's outputs) and tests/test_golden_recolour.py."""
import numpy as np

# (name, cloud kind, points, seed, scale source -> target, recolour_params keywords)
CASES = [
    ("dense_half", "dense", 6000, 3, 0.5, dict()),                 # dyadic scale: equidistant candidates at nearly every point
    ("dense_quarter_k2", "dense", 6000, 4, 0.25, dict(k_bwd=2)),
    ("dense_same", "dense", 4000, 5, 1.0, dict()),
    ("dense_generic", "dense", 6000, 6, 0.37, dict()),
    ("dense_thresholds", "dense", 6000, 7, 0.5, dict(max_attr_bwd=300.0, max_attr_fwd=200.0, skip_bwd=True)),
    ("dense_unweighted", "dense", 5000, 8, 0.37, dict(weighted_fwd=False, weighted_bwd=False)),
    ("dense_range2", "dense", 5000, 9, 0.5, dict(search_range=2)),
    ("lidar_quarter", "lidar", 6000, 10, 0.25, dict()),
    ("lidar_generic", "lidar", 6000, 11, 0.013, dict(k_bwd=3)),
    ("lidar_offsets", "lidar", 5000, 12, 0.25, dict(dist_offset_fwd=1.0, dist_offset_bwd=0.5)),
]


def make_inputs(case):
    """-> (params, source xyz, source attrs, target xyz, scale)"""
    from mpeg_pcc_tmc13_amd import recolour_params, synth
    name, kind, n, seed, scale, kw = case
    xyz, a = synth.dense_cloud(n, seed=seed, bits=9) if kind == "dense" else synth.lidar_cloud(n, seed=seed)
    tgt = np.unique(np.rint(xyz.astype(np.float64) * scale).astype(np.int32), axis=0)
    return recolour_params(bitdepth=8, **kw), xyz, a, tgt, scale

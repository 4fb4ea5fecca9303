"""estimateDist2 (tmc3/AttributeEncoder.cpp:1684-1720): oracle vs compiled
reference on the CPU, device vs both on the GPU."""
import numpy as np
import pytest

import lod_helpers as lh
import oracle_loader as ol

import conftest  # noqa: F401  (puts the package on the path)


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


CASES = [("dense", 60000, 9, 1), ("dense", 200000, 10, 2), ("lidar", 150000, 0, 3), ("random", 5000, 12, 4),
         ("random", 3, 8, 5), ("random", 1, 8, 6), ("random", 130, 6, 7)]
PARAMS = [(100, 128, 0.85), (7, 16, 0.5), (100, 128, 0.0), (13, 300, 0.99)]


def cloud(kind, n, bits, seed):
    from mpeg_pcc_tmc13_amd import synth
    if kind == "dense":
        xyz = synth.dense_cloud(n, seed=seed, bits=bits)[0]
    elif kind == "lidar":
        xyz = synth.lidar_cloud(n, seed=seed)[0]
    else:
        xyz = synth.random_cloud(n, seed=seed, bits=bits)[0]
    # estimateDist2 walks the cloud in coded (Morton) order
    codes = synth.morton_codes(xyz)
    return np.ascontiguousarray(xyz[np.lexsort((np.arange(len(xyz)), codes))])


@pytest.mark.skipif(not ol.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
def test_oracle_matches_reference(case):
    xyz = cloud(*case)
    for prm in PARAMS:
        assert lh.oracle_estimate_dist2(xyz, *prm) == lh.ref_estimate_dist2(xyz, *prm), prm


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
def test_device_matches_checkers(case, ctx):
    xyz = cloud(*case)
    for prm in PARAMS:
        want = lh.ref_estimate_dist2(xyz, *prm) if ol.ref_available() else lh.oracle_estimate_dist2(xyz, *prm)
        assert ctx.estimate_dist2(xyz, *prm) == want, prm
        assert lh.oracle_estimate_dist2(xyz, *prm) == want


@pytest.mark.gpu
def test_device_large(ctx):
    xyz = cloud("dense", 1000000, 10, 9)
    assert ctx.estimate_dist2(xyz) == lh.oracle_estimate_dist2(xyz)

"""LoD generation against the COMMITTED golden vectors of the compiled
reference (tests/golden/lod_golden.npz, generator make_lod_golden.py): the CPU
oracle on every box, the device path on the GPU box.  Neither needs the
reference at run time.  Bit-exact."""
import hashlib
import os

import numpy as np
import pytest

import conftest  # noqa: F401
import lod_helpers as lh
from lod_cases import CLOUDS, VARIANTS, make_cloud, make_params

KEYS = ("npl", "indexes", "nc", "ni", "w")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "lod_golden.npz"))


def digest(r):
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(np.asarray(r[k]).astype(np.int64)).tobytes())
    return h.hexdigest()


def check(r, golden, cname, vi):
    if f"{cname}/{vi}/npl" in golden:
        for k in KEYS:
            np.testing.assert_array_equal(np.asarray(r[k]).astype(np.int64), golden[f"{cname}/{vi}/{k}"],
                                          err_msg=f"{cname} {VARIANTS[vi]} {k}")
    assert digest(r) == str(golden[f"{cname}/{vi}/sha"]), f"{cname} {VARIANTS[vi]}"


@pytest.mark.parametrize("cname", CLOUDS)
def test_oracle_matches_golden(cname, golden):
    xyz = make_cloud(cname)
    for vi, kw in enumerate(VARIANTS):
        check(lh.oracle_lod_generate(xyz, make_params(kw)), golden, cname, vi)
    assert [lh.oracle_estimate_dist2(xyz, 100, 128, 0.85), lh.oracle_estimate_dist2(xyz, 7, 16, 0.5)] \
        == list(golden[f"{cname}/dist2"])


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cname", CLOUDS)
def test_device_matches_golden(cname, golden, ctx):
    xyz = make_cloud(cname)
    for vi, kw in enumerate(VARIANTS):
        check(ctx.lod_build(make_params(kw), xyz), golden, cname, vi)
    assert [ctx.estimate_dist2(xyz, 100, 128, 0.85), ctx.estimate_dist2(xyz, 7, 16, 0.5)] \
        == list(golden[f"{cname}/dist2"])

"""GPU parity of the lifting transform (predictors given) and of
PCCPredictor::computeWeights through the C ABI: against the committed
golden vectors of the compiled reference, the CPU oracle, and (where the
compiled reference travelled) live LoD structures of bigger clouds."""
import ast
import os

import numpy as np
import pytest

import lod_helpers as lh
import oracle_loader as ol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "lift_golden.npz")
NAMES = ["dense3k_qp34", "dense3k_qp10", "rand2k_layers", "lidar3k_refl", "tiny5"]


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", NAMES)
def test_lift_vs_golden(name, ctx):
    from mpeg_pcc_tmc13_amd import lift_params
    g = np.load(GOLDEN)
    pk = ast.literal_eval(str(g[name + "/params"]))
    attrs = g[name + "/attrs"]
    c = attrs.shape[1]
    lf = lift_params(g[name + "/npl"], lcp=(c == 3), **pk)
    co, rec, lcp = ctx.lift_forward(lf, g[name + "/nc"], g[name + "/ni"], g[name + "/w"], g[name + "/indexes"], attrs)
    np.testing.assert_array_equal(co, g[name + "/coeffs"])
    np.testing.assert_array_equal(rec, g[name + "/rec"])
    if c == 3:
        nl = len(g[name + "/npl"])
        np.testing.assert_array_equal(lcp[:nl], g[name + "/lcp"][:nl])
    inv = ctx.lift_inverse(lf, g[name + "/nc"], g[name + "/ni"], g[name + "/w"], g[name + "/indexes"],
                           g[name + "/coeffs"], lcp=g[name + "/lcp"])
    np.testing.assert_array_equal(inv, g[name + "/rec"])
    # computeWeights: raw squared distances -> the weights above
    nc2, w2 = ctx.lod_compute_weights(g[name + "/nc_raw"], g[name + "/dist2"])
    np.testing.assert_array_equal(nc2, g[name + "/nc"])
    np.testing.assert_array_equal(w2, g[name + "/w"])


@pytest.mark.skipif(not ol.ref_available(), reason="compiled reference absent")
@pytest.mark.parametrize("kind,n,c", [("dense", 200000, 3), ("lidar", 150000, 1)])
def test_lift_large_vs_oracle(kind, n, c, ctx):
    from mpeg_pcc_tmc13_amd import lift_params, lod_params, synth
    xyz, attrs = synth.dense_cloud(n, seed=31, bits=10) if kind == "dense" else synth.lidar_cloud(n, seed=31)
    lod = lh.ref_lod_generate(xyz, lod_params())
    rng = np.random.default_rng(1)
    qp_off = rng.integers(-3, 4, size=(len(xyz), 2)).astype(np.int32)
    for q in (None, qp_off):
        lf = lift_params(lod["npl"], qp=34, chroma_offset=-1 if c == 3 else 0, lcp=(c == 3),
                         layers=[(34, -1 if c == 3 else 0), (30, 0), (38, 0)])
        o_co, o_rec, o_lcp = lh.lift(ol.oracle(), True, lf, lod, attrs, qp_off=q)
        co, rec, lcp = ctx.lift_forward(lf, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"],
                                        attrs, qp_off=q)
        np.testing.assert_array_equal(co, o_co)
        np.testing.assert_array_equal(rec, o_rec)
        inv = ctx.lift_inverse(lf, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], o_co,
                               lcp=o_lcp, qp_off=q)
        np.testing.assert_array_equal(inv, o_rec)


def test_lift_rejects_bad_structure(ctx):
    from mpeg_pcc_tmc13_amd import lift_params
    from mpeg_pcc_tmc13_amd._lib import GpccError
    lf = lift_params([1, 3], qp=20)
    nc = np.array([0, 1, 1], np.int32)
    ni = np.array([[0, 0, 0], [0, 0, 0], [2, 0, 0]], np.int32)  # neighbour does not precede
    w = np.full((3, 3), 256, np.int32)
    with pytest.raises(GpccError):
        ctx.lift_forward(lf, nc, ni, w, np.arange(3, dtype=np.int32), np.zeros((3, 1), np.int32))


@pytest.mark.parametrize("kind,n,kw", [("dense", 60000, {}), ("lidar", 40000, dict(decimation=1)),
                                       ("dense", 3000, dict(decimation=2)), ("random", 5, {}), ("random", 1, {})])
def test_lift_attr_driver_matches_composition(kind, n, kw, ctx):
    """gpcc_lift_encode_attr / gpcc_lift_decode_attr (LoD build + lifting in one
    call, predictors never leave the device) == gpcc_lod_build + gpcc_lift_*
    == CPU checker on the same structure."""
    from mpeg_pcc_tmc13_amd import lift_params, lod_params, synth
    if kind == "dense":
        xyz, attrs = synth.dense_cloud(n, seed=51, bits=9)
    elif kind == "lidar":
        xyz, attrs = synth.lidar_cloud(n, seed=51)
    else:
        xyz, attrs = synth.random_cloud(n, seed=51, bits=3)
    lp = lod_params(**kw)
    g = ctx.lod_build(lp, xyz)
    lf = lift_params(g["npl"], qp=34)
    co0, rec0, lcp0 = ctx.lift_forward(lf, g["nc"], g["ni"], g["w"], g["indexes"], attrs)
    lf2 = lift_params([len(xyz)], qp=34)   # LoD sizes are produced by the call
    co, rec, lcp, idx = ctx.lift_encode_attr(lp, lf2, xyz, attrs)
    assert lf2.num_lods == len(g["npl"]) and list(lf2.num_points_in_lod[:lf2.num_lods]) == list(g["npl"])
    np.testing.assert_array_equal(idx, g["indexes"])
    np.testing.assert_array_equal(co, co0)
    np.testing.assert_array_equal(rec, rec0)
    np.testing.assert_array_equal(lcp, lcp0)
    lf3 = lift_params([len(xyz)], qp=34)
    dec = ctx.lift_decode_attr(lp, lf3, xyz, co, lcp)
    np.testing.assert_array_equal(dec, rec)
    o = lh.oracle_lod_generate(xyz, lp)
    o_co, o_rec, _ = lh.lift(ol.oracle(), True, lf, o, attrs)
    np.testing.assert_array_equal(co, o_co)
    np.testing.assert_array_equal(rec, o_rec)


@pytest.mark.parametrize("kind,n", [("dense", 60000), ("lidar", 40000), ("random", 5), ("random", 1)])
def test_scalable_lifting_attr_driver_vs_oracle(kind, n, ctx):
    """aps.scalable_lifting_enabled_flag through the one-call entries: the LoD structure of
    lod_scalable.hpp and the quantisation weights by level of detail
    (computeQuantizationWeightsScalable) == the oracle, which gives the reference operator's
    payload byte for byte (tests/test_oracle_lift.py); the decoder returns the reconstruction."""
    from mpeg_pcc_tmc13_amd import lift_params, lod_params, synth
    if kind == "dense":
        xyz, attrs = synth.dense_cloud(n, seed=52, bits=9)
    elif kind == "lidar":
        xyz, attrs = synth.lidar_cloud(n, seed=52)
    else:
        xyz, attrs = synth.random_cloud(n, seed=52, bits=3)
    c = attrs.shape[1]
    lp = lod_params()
    lp.scalable_lifting_enabled_flag = 1
    lp.max_neigh_range_minus1 = 5
    o = lh.oracle_lod_generate(xyz, lp)
    lf = lift_params(o["npl"], qp=34, lcp=(c == 3), scalable=True)
    o_co, o_rec, o_lcp = lh.lift(ol.oracle(), True, lf, o, attrs)
    lf2 = lift_params([len(xyz)], qp=34, lcp=(c == 3))   # LoD sizes and the scalable flag come from the call
    co, rec, lcp, idx = ctx.lift_encode_attr(lp, lf2, xyz, attrs)
    assert lf2.scalable_lifting_enabled_flag == 1
    assert list(lf2.num_points_in_lod[:lf2.num_lods]) == list(o["npl"])
    np.testing.assert_array_equal(idx, o["indexes"])
    np.testing.assert_array_equal(co, o_co)
    np.testing.assert_array_equal(rec, o_rec)
    if c == 3:
        np.testing.assert_array_equal(np.asarray(lcp)[:len(o["npl"])], o_lcp[:len(o["npl"])])
    lf3 = lift_params([len(xyz)], qp=34, lcp=(c == 3))
    dec = ctx.lift_decode_attr(lp, lf3, xyz, co, lcp)
    np.testing.assert_array_equal(dec, rec)
    # the two-call form: structure out, lifting with the flag set by the caller
    g = ctx.lod_build(lp, xyz)
    co2, rec2, _ = ctx.lift_forward(lf, g["nc"], g["ni"], g["w"], g["indexes"], attrs)
    np.testing.assert_array_equal(co2, o_co)
    np.testing.assert_array_equal(rec2, o_rec)

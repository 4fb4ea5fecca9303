"""Predicting transform on the device (gpcc_pred_*) against the oracle
(oracle/pred_oracle.c, pinned to the compiled reference at symbol level by
tests/test_oracle_pred.py) and, where oracle/_ref is present, against the
reference operator itself.  The decoder covers every tool (prediction modes,
inter-component prediction, QP layers, region offsets), and so does the encoder:
its choice among direct predictors (a running rate model in the reference) is
iterated to the sequential coder's fixed point."""
import numpy as np
import pytest

import conftest  # noqa: F401
import lod_helpers as lh
import oracle_loader as ol
import test_oracle_pred as top

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


def oracle_case(name):
    """-> xyz, attrs, lod (oracle LoD structure), pp, oracle (values, rec, icp, modes)"""
    from mpeg_pcc_tmc13_amd import pred_params
    xyz, attrs, lp, qp, bitdepth, thr, po = top.make(name)
    lod = lh.oracle_lod_generate(xyz, lp)
    pp = pred_params(lod["npl"], qp=qp, chroma_offset=0, bitdepth=bitdepth, threshold=thr,
                     max_levels=lp.num_detail_levels_minus1 + 1, **po)
    return xyz, attrs, lp, lod, pp, lh.oracle_pred(True, pp, lod, attrs=attrs)


@pytest.mark.parametrize("name", list(top.CASES))
def test_decoder_matches_oracle(name, ctx):
    xyz, attrs, lp, lod, pp, (values, rec, icp, modes) = oracle_case(name)
    got = ctx.pred_inverse(pp, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], values, icp=icp)
    np.testing.assert_array_equal(got, rec)
    # one call from positions: LoD structure built and consumed on the device
    from mpeg_pcc_tmc13_amd import pred_params
    pp2 = pred_params([len(xyz)], qp=pp.layer_qp[0][0], chroma_offset=0, bitdepth=pp.bitdepth,
                      direct=pp.max_num_direct_predictors, avg_disabled=bool(pp.direct_avg_predictor_disabled_flag),
                      threshold=pp.adaptive_prediction_threshold >> max(0, pp.bitdepth - 8),
                      icp=bool(pp.inter_component_prediction_enabled_flag),
                      quant_neigh_weight=tuple(pp.quant_neigh_weight), max_levels=pp.max_num_detail_levels)
    got2 = ctx.pred_decode_attr(lp, pp2, xyz, values, icp=icp)
    np.testing.assert_array_equal(got2, rec)
    assert list(pp2.num_points_in_lod[:pp2.num_lods]) == list(lod["npl"])


@pytest.mark.skipif(not (ol.ref_available() and lh.entropy_dec_available()), reason="compiled reference absent")
@pytest.mark.parametrize("name", ["dense_ctc", "lidar_refl_ctc", "dense_avg_disabled", "dense_qnw_direct"])
def test_decoder_from_reference_bitstream_symbols(name, ctx):
    xyz, attrs, lod, pp, values, want_rec, icp = top.run_reference(name)
    got = ctx.pred_inverse(pp, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], values, icp=icp)
    np.testing.assert_array_equal(got, want_rec)


ENC_CASES = ["dense_nodirect_qnw", "tiny", "single"]


@pytest.mark.parametrize("name", ENC_CASES + ["dense_ctc", "lidar_refl_lods", "dense_skip_intra", "random_sparse"])
def test_encoder_without_direct_predictors_matches_oracle(name, ctx):
    from mpeg_pcc_tmc13_amd import pred_params
    xyz, attrs, lp, qp, bitdepth, thr, po = top.make(name)
    po = dict(po, direct=0)
    lod = lh.oracle_lod_generate(xyz, lp)
    pp = pred_params(lod["npl"], qp=qp, chroma_offset=0, bitdepth=bitdepth, threshold=thr,
                     max_levels=lp.num_detail_levels_minus1 + 1, **po)
    want_v, want_rec, want_icp, _ = lh.oracle_pred(True, pp, lod, attrs=attrs)
    v, rec, icp = ctx.pred_forward(pp, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], attrs)
    if attrs.shape[1] == 3 and pp.inter_component_prediction_enabled_flag:
        np.testing.assert_array_equal(icp, want_icp)
    np.testing.assert_array_equal(v, want_v)
    np.testing.assert_array_equal(rec, want_rec)
    # one call from positions, then the device decoder gives the encoder's reconstruction back
    pp2 = pred_params([len(xyz)], qp=qp, chroma_offset=0, bitdepth=bitdepth, threshold=thr,
                      max_levels=lp.num_detail_levels_minus1 + 1, **po)
    v2, rec2, icp2, idx = ctx.pred_encode_attr(lp, pp2, xyz, attrs)
    np.testing.assert_array_equal(v2, want_v)
    np.testing.assert_array_equal(rec2, want_rec)
    np.testing.assert_array_equal(idx, lod["indexes"])
    np.testing.assert_array_equal(ctx.pred_decode_attr(lp, pp2, xyz, v2, icp=icp2), want_rec)


@pytest.mark.parametrize("name", list(top.CASES))
def test_encoder_with_direct_predictors_matches_oracle(name, ctx):
    """the CTC encoder (three direct predictors, AttributeEncoder.cpp:663-745, 896-985): the
    choice among them reads the running rate model; the device iterates the DAG pass and the
    model's trajectory to their fixed point, which IS the sequential coder's result -- values
    (mode bits included), reconstruction and inter-component coefficients equal the oracle's
    (pinned to the reference's own bitstream symbols by tests/test_oracle_pred.py)"""
    xyz, attrs, lp, lod, pp, (want_v, want_rec, want_icp, modes) = oracle_case(name)
    before = ctx.stats()
    v, rec, icp = ctx.pred_forward(pp, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], attrs)
    if attrs.shape[1] == 3 and pp.inter_component_prediction_enabled_flag:
        np.testing.assert_array_equal(icp, want_icp)
    np.testing.assert_array_equal(v, want_v)
    np.testing.assert_array_equal(rec, want_rec)
    assert ctx.stats()["calls_unsupported"] == before["calls_unsupported"]
    # and the decoder gives the encoder's reconstruction back
    np.testing.assert_array_equal(
        ctx.pred_inverse(pp, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], v, icp=icp), rec)


def test_one_call_encoders_take_the_ctc_configuration(ctx):
    """gpcc_pred_encode_attr / gpcc_dev_pred_encode_attr with three direct predictors (the
    reference's default; declined until round 3): same values as the oracle's encoder on the
    oracle's LoD structure"""
    import torch
    from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth
    xyz, attrs = synth.dense_cloud(20000, seed=3, bits=8)
    lp = lod_params()
    lod = lh.oracle_lod_generate(xyz, lp)
    ppo = pred_params(lod["npl"], qp=34, direct=3, max_levels=lp.num_detail_levels_minus1 + 1)
    want_v, want_rec, want_icp, _ = lh.oracle_pred(True, ppo, lod, attrs=attrs)
    pp = pred_params([len(xyz)], qp=34, direct=3, max_levels=lp.num_detail_levels_minus1 + 1)
    v, rec, icp, idx = ctx.pred_encode_attr(lp, pp, xyz, attrs)
    np.testing.assert_array_equal(v, want_v)
    np.testing.assert_array_equal(rec, want_rec)
    dev = torch.device("cuda:0")
    d_xyz = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev)
    d_a = torch.from_numpy(np.ascontiguousarray(attrs).reshape(-1)).to(dev)
    d_v = torch.zeros_like(d_a)
    pp3 = pred_params([len(xyz)], qp=34, direct=3, max_levels=lp.num_detail_levels_minus1 + 1)
    ctx.dev_pred_attr(True, lp, [pp3], [0, len(xyz)], d_xyz.data_ptr(), d_a.data_ptr(), d_v.data_ptr(), 3)
    ctx.synchronize()
    np.testing.assert_array_equal(d_v.cpu().numpy().reshape(-1, 3), want_v)


def test_qp_layers_and_region_offsets(ctx):
    from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth
    xyz, attrs = synth.dense_cloud(20000, seed=5, bits=8)
    lp = lod_params(levels=8, lifting=False, intra_range=1100000, blend=True)
    lp.intra_lod_prediction_skip_layers = 0
    lod = lh.oracle_lod_generate(xyz, lp)
    layers = [(28 + 2 * i, -1 if i & 1 else 1) for i in range(5)]
    pp = pred_params(lod["npl"], bitdepth=8, layers=layers, max_levels=8, quant_neigh_weight=(25, 12, 12))
    rng = np.random.default_rng(3)
    qp_off = np.zeros((len(xyz), 2), np.int32)
    sel = xyz[:, 0] < xyz[:, 0].mean()
    qp_off[sel] = (4, -2)
    qp_off[rng.random(len(xyz)) < 0.05] = (-30, 9)  # drives the clip to [4, maxQp]
    values, rec, icp, modes = lh.oracle_pred(True, pp, lod, attrs=attrs, qp_off=qp_off)
    assert (modes > 0).any()
    got = ctx.pred_inverse(pp, lod["nc"], lod["ni"], lod["w"].astype(np.int32), lod["indexes"], values, icp=icp,
                           qp_off=qp_off)
    np.testing.assert_array_equal(got, rec)


@pytest.mark.parametrize("kind,n,levels", [("dense", 300000, 12), ("lidar", 120000, 1), ("lidar", 300000, 10)])
def test_decoder_larger_slices(kind, n, levels, ctx):
    """a 12-LoD dense slice, a single-LoD scan-ordered LiDAR slice (cat3 CTC:
    the whole slice is one level of detail, the DAG is deep) and a LiDAR slice
    with LoDs"""
    from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth
    xyz, attrs = synth.dense_cloud(n, seed=9, bits=10) if kind == "dense" else synth.lidar_cloud(n, seed=9)
    c = attrs.shape[1]
    lp = lod_params(levels=levels, lifting=False, intra_range=1100000, blend=(c == 3))
    lp.intra_lod_prediction_skip_layers = 0
    lod = ctx.lod_build(lp, xyz)
    pp = pred_params(lod["npl"], qp=28, bitdepth=8 if c == 3 else 16, max_levels=levels,
                     avg_disabled=(levels == 1))
    values, rec, icp, _ = lh.oracle_pred(True, pp, lod, attrs=attrs)
    got = ctx.pred_inverse(pp, lod["nc"], lod["ni"], lod["w"], lod["indexes"], values, icp=icp)
    np.testing.assert_array_equal(got, rec)


@pytest.mark.skipif(not (ol.ref_available() and lh.entropy_available()), reason="compiled reference absent")
@pytest.mark.parametrize("name", ["dense_nodirect_qnw", "dense_ctc", "lidar_refl_lods"])
def test_device_encoder_bitstream_identical_to_reference_operator(name, ctx):
    """positions + attributes -> device LoD build + predicting transform (no
    direct predictors) -> device zero-run packing -> the reference's own
    arithmetic coder: the bytes equal the arithmetic-coded part of the payload
    AttributeEncoder::encode writes for the slice."""
    from mpeg_pcc_tmc13_amd import pred_params
    xyz, attrs, lp, qp, bitdepth, thr, po = top.make(name)
    po = dict(po, direct=0)
    n, c = attrs.shape
    pp = pred_params([n], qp=qp, chroma_offset=0, bitdepth=bitdepth, threshold=thr,
                     max_levels=lp.num_detail_levels_minus1 + 1, **po)
    payload, rec_enc, rec_dec, want_icp = lh.ref_pred_roundtrip(lp, pp, thr, qp, 0, xyz, attrs)
    np.testing.assert_array_equal(rec_enc, rec_dec)
    v, rec, icp, idx = ctx.pred_encode_attr(lp, pp, xyz, attrs)
    np.testing.assert_array_equal(rec, rec_enc)
    if c == 3:
        np.testing.assert_array_equal(icp, want_icp)
    runs, vals, trailing = ctx.zero_run_pack(v, n, c, planar=False)
    assert lh.ref_entropy_encode_symbols(c, n, runs, vals, trailing) == payload[lh.ref_last_abh_size():]


def test_device_tier_equals_host_tier_over_ragged_slices():
    """gpcc_dev_pred_encode_attr / _decode_attr on ragged slices resident in HBM
    (concurrent lanes) == the host-tier one-call entries slice by slice; the
    decoder also takes the CTC stream with direct predictors from the oracle."""
    import torch
    from mpeg_pcc_tmc13_amd import context, lod_params, pred_params, synth
    ctx = context(0)
    dev = torch.device("cuda:0")
    sizes = [20_000, 3, 60_000, 1, 9_000, 31_000]
    clouds = [synth.dense_cloud(n, seed=700 + i, bits=9 if n > 1000 else 4) for i, n in enumerate(sizes)]
    sizes = [len(c[0]) for c in clouds]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offsets[-1])
    lp = lod_params(levels=10, lifting=False, intra_range=1100000, blend=True)
    lp.intra_lod_prediction_skip_layers = 0
    kw = dict(qp=31, bitdepth=8, max_levels=10, quant_neigh_weight=(16, 8, 4))
    d_xyz = torch.from_numpy(np.concatenate([c[0] for c in clouds])).to(dev)
    d_attrs = torch.from_numpy(np.concatenate([c[1] for c in clouds]).reshape(-1)).to(dev)
    d_vals = torch.zeros(3 * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.set_morton_bits(27)
    pps = [pred_params([s], direct=0, **kw) for s in sizes]
    icp = ctx.dev_pred_attr(True, lp, pps, offsets, d_xyz.data_ptr(), d_attrs.data_ptr(), d_vals.data_ptr(), 3)
    rec, vals = d_attrs.cpu().numpy().reshape(-1, 3), d_vals.cpu().numpy().reshape(-1, 3)
    d_dec = torch.zeros(3 * n, dtype=torch.int32, device=dev)
    pps2 = [pred_params([s], direct=0, **kw) for s in sizes]
    ctx.dev_pred_attr(False, lp, pps2, offsets, d_xyz.data_ptr(), d_dec.data_ptr(), d_vals.data_ptr(), 3, icp=icp)
    np.testing.assert_array_equal(d_dec.cpu().numpy().reshape(-1, 3), rec)
    ctc_vals, ctc_rec, ctc_icp = [], [], np.zeros((len(sizes), 32, 3), np.int8)
    for i, (xyz, attrs) in enumerate(clouds):
        a, b = int(offsets[i]), int(offsets[i + 1])
        pp = pred_params([len(xyz)], direct=0, **kw)
        v, r, l, idx = ctx.pred_encode_attr(lp, pp, xyz, attrs)
        np.testing.assert_array_equal(vals[a:b], v)
        np.testing.assert_array_equal(rec[a:b], r)
        np.testing.assert_array_equal(icp[i], l)
        assert list(pps[i].num_points_in_lod[:pps[i].num_lods]) == list(pp.num_points_in_lod[:pp.num_lods])
        # the reference's stream for the same slice with three direct predictors (oracle encoder)
        lod = lh.oracle_lod_generate(xyz, lp)
        ppc = pred_params(lod["npl"], direct=3, **kw)
        ov, orec, oicp, _ = lh.oracle_pred(True, ppc, lod, attrs=attrs)
        ctc_vals.append(ov)
        ctc_rec.append(orec)
        ctc_icp[i] = oicp
    d_v2 = torch.from_numpy(np.concatenate(ctc_vals).reshape(-1)).to(dev)
    torch.cuda.synchronize()
    pps3 = [pred_params([s], direct=3, **kw) for s in sizes]
    ctx.dev_pred_attr(False, lp, pps3, offsets, d_xyz.data_ptr(), d_dec.data_ptr(), d_v2.data_ptr(), 3, icp=ctc_icp)
    np.testing.assert_array_equal(d_dec.cpu().numpy().reshape(-1, 3), np.concatenate(ctc_rec))
    ctx.close()

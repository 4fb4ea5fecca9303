"""CPU-only: the LoD build of scalable lifting as the LIBRARY runs it -- the level loop of
lod_scalable.hpp and the kernels of lod_kernels.hpp, compiled for the CPU wavefront emulator
(tests/emu) -- and the neighbour search with attribute inter prediction
(lod_nn_search_kernel<false, true>, lod_finalise_inter_kernel, lod_blend_weights_inter_kernel)
against the oracle (oracle/lod_oracle.c, pinned to the compiled reference by
tests/test_oracle_lod.py).  Bit-exact."""
import numpy as np
import pytest

import emu_lod_loader as el
import lod_helpers as lh


def clouds():
    from mpeg_pcc_tmc13_amd import synth
    return [("rand5", synth.random_cloud(5, seed=24, bits=2)[0]),
            ("one", synth.random_cloud(1, seed=1, bits=3)[0]),
            ("two", synth.random_cloud(2, seed=1, bits=3)[0]),
            ("rand3k", synth.random_cloud(3000, seed=2, bits=5)[0]),
            ("dups", synth.random_cloud(400, seed=9, bits=2, dup_fraction=0.3)[0]),
            ("dense9k", synth.dense_cloud(9000, seed=4, bits=7)[0]),
            ("lidar7k", synth.lidar_cloud(7000, seed=3)[0]),
            ("sparse", synth.random_cloud(2000, seed=8, bits=20)[0])]


VARIANTS = [dict(), dict(bias=(1, 2, 1)), dict(neighbours=2), dict(distribution=False), dict(intra_range=16),
            dict(inter_range=8), dict(lifting=False, intra_range=64, blend=True)]


@pytest.mark.parametrize("vi", range(len(VARIANTS)))
def test_scalable_lod_build_under_the_emulator(vi):
    from mpeg_pcc_tmc13_amd import lod_params
    kw = VARIANTS[vi]
    for name, xyz in clouds():
        for rng in (0, 4, 40):
            lp = lod_params(**kw)
            if kw.get("lifting") is False:
                lp.intra_lod_prediction_skip_layers = 0
            lp.scalable_lifting_enabled_flag = 1
            lp.max_neigh_range_minus1 = rng
            el.assert_same_lod(el.scalable_build(lp, xyz), lh.oracle_lod_generate(xyz, lp), f"{name} {kw} range={rng}")


def test_scalable_lod_build_larger_cloud():
    from mpeg_pcc_tmc13_amd import lod_params, synth
    for xyz in (synth.dense_cloud(60000, seed=14, bits=9)[0], synth.lidar_cloud(40000, seed=13)[0]):
        lp = lod_params()
        lp.scalable_lifting_enabled_flag = 1
        lp.max_neigh_range_minus1 = 5
        el.assert_same_lod(el.scalable_build(lp, xyz), lh.oracle_lod_generate(xyz, lp))


INTER = [dict(decimation=1), dict(decimation=2), dict(decimation=1, distribution=False), dict(decimation=2, bias=(1, 2, 1)),
         dict(decimation=1, neighbours=2), dict(decimation=2, lifting=False, intra_range=64, blend=True),
         dict(decimation=1, levels=1), dict(decimation=2, sampling_period=3, dist2=1),
         dict(), dict(dist2=1, distribution=False), dict(lifting=False, intra_range=64, blend=True)]   # (the distance sub-sampler)


@pytest.mark.parametrize("vi", range(len(INTER)))
def test_inter_frame_search_under_the_emulator(vi):
    """gpcc_lod_build_inter's kernels: candidates of the reference frame (atlas of the first 8^3
    block, window in the frame's Morton order, no duplicate tests), tagged in their index, sorted
    and replaced with the others; flags, reference-frame point indices, the frame distance in the
    weights; blendWeights over both frames."""
    from mpeg_pcc_tmc13_amd import lod_params
    kw = INTER[vi]
    rng = np.random.default_rng(5)
    refs = 0
    for name, xyz in clouds():
        keep = rng.random(len(xyz)) > 0.1
        frame = np.clip(xyz + rng.integers(-2, 3, size=xyz.shape), 0, None)[keep].astype(np.int32) if len(xyz) > 3 else xyz.copy()
        for search_range in (0, 5, 128):
            lp = lod_params(**kw)
            if kw.get("lifting") is False:
                lp.intra_lod_prediction_skip_layers = 0
            o = lh.oracle_lod_generate_inter(xyz, frame, lp, search_range, 2)
            e = el.inter_build(lp, xyz, frame, search_range, 2)
            for k in ("npl", "indexes", "nc", "ni", "ref"):
                np.testing.assert_array_equal(e[k], o[k], err_msg=f"{name} {kw} range={search_range} {k}")
            np.testing.assert_array_equal(e["w"].astype(np.uint32), (o["w"] & 0xffffffff).astype(np.uint32),
                                          err_msg=f"{name} {kw} range={search_range} w")
            refs += int(o["ref"].sum())
    assert refs > 1000


@pytest.mark.parametrize("qp", [10, 34])
def test_inter_frame_lifting_arrangement_under_the_emulator(qp):
    """gpcc_lift_forward_inter / _inverse_inter keep the lifting kernels as they are: the reference
    frame's reflectances sit behind the n working values and a flagged neighbour points there, so
    the prediction reads the frame and what the update step and the quantisation weights would have
    skipped lands in entries nobody reads.  That arrangement, with the library's kernels under the
    emulator == the oracle (pinned to the reference operator's payload, tests/test_oracle_lift.py)."""
    import oracle_loader as ol
    from mpeg_pcc_tmc13_amd import lift_params, lod_params, synth
    rng = np.random.default_rng(7)
    for xyz, attrs in (synth.lidar_cloud(9000, seed=61), synth.dense_cloud(6000, seed=3, bits=7), synth.random_cloud(5, seed=2, bits=3)):
        attrs = attrs[:, :1].copy()
        if attrs.max() > 255:
            attrs = attrs >> 8
        keep = rng.random(len(xyz)) > 0.1 if len(xyz) > 5 else np.ones(len(xyz), bool)
        xr = np.clip(xyz + rng.integers(-2, 3, size=xyz.shape), 0, None)[keep].astype(np.int32)
        ar = np.clip(attrs + rng.integers(-6, 7, size=attrs.shape), 0, 255)[keep].astype(np.int32)
        lp = lod_params()
        lod = lh.oracle_lod_generate_inter(xyz, xr, lp, 64, 1)
        lf = lift_params(lod["npl"], qp=qp, chroma_offset=0, lcp=False, bitdepth=8)
        co, rec = lh.lift_inter(ol.oracle(), True, lf, lod, attrs, ar)
        eco, erec = el.lift_inter(True, lf, lod, ar, attrs=attrs)
        np.testing.assert_array_equal(eco, co)
        np.testing.assert_array_equal(erec, rec)
        _, einv = el.lift_inter(False, lf, lod, ar, coeffs=co)
        np.testing.assert_array_equal(einv, rec)


def test_inter_frame_predicting_transform_under_the_emulator():
    """gpcc_pred_forward_inter / _inverse_inter: the DAG pass in its inter build (a neighbour index >= n
    names the reference frame's entry, which never has to be waited for) and the share arrays with spare
    entries behind the predictors, under the emulator: the decoder with every tool combination (direct
    predictors chosen in the reference frame included), the encoder -- with direct predictors the
    pass and its rate model iterated to the fixed point --, the quantisation weights with neighbour shares
    == the oracle (pinned to the reference operator's bitstream symbols, tests/test_oracle_pred.py)."""
    from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth
    rng = np.random.default_rng(7)
    for xyz, attrs in (synth.lidar_cloud(1500, seed=61), synth.dense_cloud(1500, seed=3, bits=7), synth.random_cloud(5, seed=2, bits=3)):
        attrs = attrs[:, :1].copy()
        if attrs.max() > 255:
            attrs = attrs >> 8
        keep = rng.random(len(xyz)) > 0.1 if len(xyz) > 5 else np.ones(len(xyz), bool)
        xr = np.clip(xyz + rng.integers(-2, 3, size=xyz.shape), 0, None)[keep].astype(np.int32)
        ar = np.clip(attrs + rng.integers(-6, 7, size=attrs.shape), 0, 255)[keep].astype(np.int32)
        lp = lod_params(lifting=False, intra_range=64)
        lp.intra_lod_prediction_skip_layers = 0
        lod = lh.oracle_lod_generate_inter(xyz, xr, lp, 64, 1)
        for direct, qp, qnw in ((3, 4, (0, 0, 0)), (3, 28, (0, 0, 0)), (0, 16, (0, 0, 0)), (0, 16, (25, 12, 12)), (1, 10, (25, 12, 12))):
            pp = pred_params(lod["npl"], qp=qp, chroma_offset=0, bitdepth=8, threshold=4, direct=direct, icp=False,
                             quant_neigh_weight=qnw, max_levels=lp.num_detail_levels_minus1 + 1)
            v, rec, modes = lh.pred_inter(True, pp, lod, ar, attrs=attrs)
            _, dec = el.pred_inter(False, pp, lod, ar, values=v)
            np.testing.assert_array_equal(dec, rec, err_msg=f"decoder direct={direct} qp={qp} qnw={qnw}")
            # the encoder: with direct predictors the DAG pass and the rate model's trajectory iterated
            # to their fixed point, as the library does
            ev, erec = el.pred_inter(True, pp, lod, ar, attrs=attrs)
            np.testing.assert_array_equal(ev, v, err_msg=f"encoder values direct={direct} qp={qp} qnw={qnw}")
            np.testing.assert_array_equal(erec, rec)


def test_intra_predicting_transform_kernels_under_the_emulator():
    """The same kernels on an intra structure (no neighbour flagged): the quantisation-weight kernel with
    neighbour shares -- packed count / sum words, LDS slots inside a claim -- and the DAG pass, decoder and
    encoder without direct predictors, == the oracle."""
    from mpeg_pcc_tmc13_amd import lod_params, pred_params, synth
    for xyz, attrs in (synth.lidar_cloud(2500, seed=5), synth.dense_cloud(3000, seed=3, bits=7)):
        attrs = attrs[:, :1].copy()
        if attrs.max() > 255:
            attrs = attrs >> 8
        lp = lod_params(lifting=False, intra_range=64)
        lp.intra_lod_prediction_skip_layers = 0
        lod = dict(lh.oracle_lod_generate(xyz, lp))
        lod["ref"] = np.zeros((len(xyz), 3), np.int32)
        for direct, qnw in ((0, (25, 12, 12)), (0, (120, 80, 60)), (3, (16, 8, 4))):
            pp = pred_params(lod["npl"], qp=16, chroma_offset=0, bitdepth=8, threshold=4, direct=direct, icp=False,
                             quant_neigh_weight=qnw, max_levels=lp.num_detail_levels_minus1 + 1)
            v, rec, _, modes = lh.oracle_pred(True, pp, lod, attrs=attrs)
            _, dec = el.pred_inter(False, pp, lod, attrs[:1], values=v)
            np.testing.assert_array_equal(dec, rec, err_msg=f"decoder direct={direct} qnw={qnw}")
            if direct == 0:
                ev, erec = el.pred_inter(True, pp, lod, attrs[:1], attrs=attrs)
                np.testing.assert_array_equal(ev, v)
                np.testing.assert_array_equal(erec, rec)


INTRA = [dict(), dict(dist2=1), dict(distribution=False), dict(decimation=1), dict(decimation=2),
         dict(lifting=False, intra_range=64, blend=True), dict(bias=(1, 2, 1)), dict(levels=3), dict(inter_range=8), dict(neighbours=2)]


@pytest.mark.parametrize("vi", range(len(INTRA)))
def test_ordinary_lod_build_under_the_emulator(vi):
    """The default LoD build as the library's kernels run it, on the CPU: the distance sub-sampler -- a
    dependency-ordered kernel whose workgroups wait for one another through tickets and 16-byte
    mail-box granules, run here with its eight workgroups alive together --, the periodic and the
    centroid ones, the neighbour search, finalise, weights, blending == the oracle."""
    from mpeg_pcc_tmc13_amd import lod_params
    kw = INTRA[vi]
    for name, xyz in clouds():
        lp = lod_params(**kw)
        if kw.get("lifting") is False:
            lp.intra_lod_prediction_skip_layers = 0
        o = lh.oracle_lod_generate(xyz, lp)
        e = el.intra_build(lp, xyz)
        for k in ("npl", "indexes", "nc", "ni"):
            np.testing.assert_array_equal(e[k], o[k], err_msg=f"{name} {kw} {k}")
        np.testing.assert_array_equal(e["w"].astype(np.uint32), (o["w"] & 0xffffffff).astype(np.uint32), err_msg=f"{name} {kw} w")


def _rate_states(up, x0=1 << 19):
    """state in front of every event, and behind the last: resStatUpdate's recurrence (AttributeEncoder.cpp:137-165)"""
    out = np.empty(len(up) + 1, np.int64)
    x = x0
    for i, u in enumerate(up.tolist()):
        out[i] = x
        x = x + (((1 << 20) - x) >> 6) if u else x - (x >> 6)
    out[len(up)] = x
    return out


@pytest.mark.parametrize("m,kind", [(0, "values"), (1, "values"), (255, "events"), (256, "values"), (257, "events"),
                                    (16_384, "values"), (16_385, "events"), (50_000, "values"), (70_001, "events"),
                                    (40_000, "runs"), (40_000, "rare")])
def test_rate_model_scan_under_the_emulator(m, kind):
    """pred_rate_scan_kernel (a wavefront packs its events' flags into LDS words, the threads run their chunks'
    recurrences from the two extreme states over a 1024-event warm-up, the states leave through LDS): against the
    recurrence run from the first event, for event counts around the chunk (256) and wavefront (16 384) sizes, dense,
    in long runs and rare events, from strided values (probResGt0) and from event bytes with the count in memory
    (probResGt1)."""
    import ctypes as C
    l = el.lib()
    rng = np.random.default_rng(m * 7 + len(kind))
    if kind == "runs":
        up = (np.arange(m) // 3000) % 2 == 0
    elif kind == "rare":
        up = rng.random(m) < 0.002
    else:
        up = rng.random(m) < 0.4
    want = _rate_states(up)
    fn = l.lod_emu_rate_scan
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    if kind == "events":
        # event bytes, capacity larger than the count (the library sizes the grid by n, the count comes from memory)
        cap = m + 5000
        ev = np.zeros(cap, np.uint8)
        ev[:m] = up
        state = np.full(cap + 1, -1, np.int32)
        assert fn(None, 0, ev.ctypes.data, cap, m, state.ctypes.data, 1, 1) == 0
        np.testing.assert_array_equal(state[:m + 1], want)
        assert (state[m + 1:] == -1).all()
    else:
        stride = 3
        vals = np.zeros((max(m, 1), stride), np.int32)
        vals[:m, 1] = np.where(up, rng.integers(1, 9, m) * rng.choice([-1, 1], m), 0)
        state = np.full((max(m, 1) + 1, 6), -1, np.int32)
        assert fn(vals[:, 1:].ctypes.data, stride, None, m, -1, state[:, 2:].ctypes.data, 6, 0) == 0
        np.testing.assert_array_equal(state[:m, 2], want[:m])
        assert (state[:, [0, 1, 3, 4, 5]] == -1).all() and (state[m:, 2] == -1).all()

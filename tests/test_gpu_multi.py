"""gpcc_multi_*: slices of one batch sharded over the devices of a node from a
single host process, results gathered on the first device.  On a one-GPU box
the same physical device is listed several times: the sharding, the concurrent
enqueue on one stream per context and the gather (a device copy instead of
RCCL send / receive) are exercised; with distinct devices the gather goes
through RCCL (checked when the box has more than one GPU)."""
import numpy as np
import pytest

import oracle_loader as ol

pytestmark = pytest.mark.gpu


def make_batch(sizes, c):
    from mpeg_pcc_tmc13_amd import synth
    ms, as_ = [], []
    for i, n in enumerate(sizes):
        xyz, a = synth.random_cloud(n, seed=600 + i, bits=6 if n > 500 else 3, c=c, dup_fraction=0.05 if n > 10 else 0.0)
        m, a, _ = synth.sort_by_morton(xyz, a)
        ms.append(m)
        as_.append(a)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return ms, as_, offsets


@pytest.mark.parametrize("ndev", [1, 2, 4, 7])
@pytest.mark.parametrize("subnode", [False, True])
def test_sharded_batch_equals_per_slice_reference(ndev, subnode):
    from mpeg_pcc_tmc13_amd import raht_params
    from mpeg_pcc_tmc13_amd.raht import MultiContext
    sizes = [20_000, 3, 9_000, 41_000, 1, 15_000]
    c = 3
    ms, as_, offsets = make_batch(sizes, c)
    p = raht_params(qp=30, subnode=subnode)
    mc = MultiContext([0] * ndev)   # more devices than slices is allowed too (ndev = 7)
    assert not mc.uses_rccl()
    co, rec = mc.raht_forward(p, offsets, np.concatenate(ms), np.concatenate(as_))
    inv = mc.raht_inverse(p, offsets, np.concatenate(ms), co, c)
    mc.close()
    o = ol.oracle()
    for i, n in enumerate(sizes):
        a, b = int(offsets[i]), int(offsets[i + 1])
        o_co, o_rec = o.raht_forward(p, ms[i], as_[i])
        np.testing.assert_array_equal(co[c * a:c * b], o_co, err_msg=f"slice {i}")
        np.testing.assert_array_equal(rec[a:b], o_rec, err_msg=f"slice {i}")
        np.testing.assert_array_equal(inv[a:b], o_rec, err_msg=f"slice {i}")


def test_distinct_devices_gather_through_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    from mpeg_pcc_tmc13_amd import raht_params
    from mpeg_pcc_tmc13_amd.raht import MultiContext
    nd = torch.cuda.device_count()
    sizes = [30_000] * (2 * nd)
    ms, as_, offsets = make_batch(sizes, 1)
    p = raht_params(qp=34, subnode=True)
    mc = MultiContext(list(range(nd)))
    assert mc.uses_rccl()
    co, rec = mc.raht_forward(p, offsets, np.concatenate(ms), np.concatenate(as_))
    mc.close()
    o = ol.oracle()
    for i, n in enumerate(sizes):
        a, b = int(offsets[i]), int(offsets[i + 1])
        o_co, o_rec = o.raht_forward(p, ms[i], as_[i])
        np.testing.assert_array_equal(co[a:b], o_co)
        np.testing.assert_array_equal(rec[a:b], o_rec)


def test_rccl_transport_selftest():
    """What a one-GPU box can check of the gather's transport: librccl loads, the seven
    symbols the library uses resolve, and a one-rank communicator moves a buffer through
    ncclSend / ncclRecv (the calls multi_transform issues) bit for bit."""
    from mpeg_pcc_tmc13_amd import _lib
    lib = _lib.load()
    lib.gpcc_multi_rccl_selftest.argtypes = [__import__("ctypes").c_int32]
    rc = lib.gpcc_multi_rccl_selftest(0)
    assert rc == 0, lib.gpcc_last_error()


def test_failed_call_leaves_the_callers_buffers(tmp_path):
    """outputs are written only once every device has finished without an error: an
    invalid batch (unsorted slice) returns an error and attrs / coeffs are untouched"""
    from mpeg_pcc_tmc13_amd import _lib, raht_params
    from mpeg_pcc_tmc13_amd.raht import MultiContext
    ms, as_, offsets = make_batch([5000, 4000], 1)
    morton = np.concatenate(ms)
    attrs = np.concatenate(as_)
    bad_offsets = offsets.copy()
    bad_offsets[1] = bad_offsets[2]  # an empty slice
    mc = MultiContext([0, 0])
    with pytest.raises(_lib.GpccError):
        mc.raht_forward(raht_params(qp=30), bad_offsets, morton, attrs)
    mc.close()


@pytest.mark.parametrize("ndev", [1, 3, 7])
@pytest.mark.parametrize("predicting", [False, True])
def test_lod_coders_sharded_equal_single_context(ndev, predicting):
    """gpcc_multi_lift_* / gpcc_multi_pred_*: configs[2]'s shape -- five ragged slices of a
    LoD-based coder sharded over the listed devices (one host thread each) -- against the
    one-call entries on one context slice by slice, and (lifting) against the oracle."""
    import ctypes as C
    from mpeg_pcc_tmc13_amd import context, lift_params, lod_params, pred_params, synth
    from mpeg_pcc_tmc13_amd.raht import MultiContext
    sizes = [9_000, 1, 14_000, 700, 6_000]
    c = 3
    clouds = [synth.dense_cloud(n, seed=910 + i, bits=7) if n > 10 else synth.random_cloud(n, seed=910 + i, bits=4, c=3)
              for i, n in enumerate(sizes)]
    xyz = np.concatenate([x for x, _ in clouds]).astype(np.int32)
    attrs = np.concatenate([a for _, a in clouds]).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    lp = lod_params(levels=8, lifting=not predicting)
    if predicting:
        lp.intra_lod_prediction_skip_layers = 0

    def block():
        return pred_params([0], qp=28, max_levels=8) if predicting else lift_params([0], qp=34, chroma_offset=-1)

    blocks = [block() for _ in sizes]
    mc = MultiContext([0] * ndev)
    v, rec, side, idx = mc.lod_encode_attr(predicting, lp, blocks, offsets, xyz, attrs)
    dec = mc.lod_decode_attr(predicting, lp, [block() for _ in sizes], offsets, xyz, v, side)
    mc.close()
    np.testing.assert_array_equal(dec, rec)
    ctx = context(0)
    for i, n in enumerate(sizes):
        a, b = int(offsets[i]), int(offsets[i + 1])
        one = block()
        if predicting:
            v1, r1, s1, i1 = ctx.pred_encode_attr(lp, one, xyz[a:b], attrs[a:b])
        else:
            v1, r1, s1, i1 = ctx.lift_encode_attr(lp, one, xyz[a:b], attrs[a:b])
        np.testing.assert_array_equal(v[a:b], v1, err_msg=f"slice {i}")
        np.testing.assert_array_equal(rec[a:b], r1, err_msg=f"slice {i}")
        np.testing.assert_array_equal(idx[a:b], i1, err_msg=f"slice {i}")
        np.testing.assert_array_equal(side[i], s1, err_msg=f"slice {i}")
        assert blocks[i].num_lods == one.num_lods
        assert list(blocks[i].num_points_in_lod[:one.num_lods]) == list(one.num_points_in_lod[:one.num_lods])


def test_lod_coders_failed_slice_fails_the_call():
    from mpeg_pcc_tmc13_amd import lift_params, lod_params
    from mpeg_pcc_tmc13_amd._lib import GpccError
    from mpeg_pcc_tmc13_amd.raht import MultiContext
    from mpeg_pcc_tmc13_amd import synth
    xyz, attrs = synth.dense_cloud(4000, seed=3, bits=6)
    offsets = np.array([0, 1500, 4000], dtype=np.int64)
    lp = lod_params(levels=8)
    lp.canonical_point_order_flag = 1  # the points are not in Morton order: declined by the device path
    mc = MultiContext([0, 0])
    with pytest.raises(GpccError) as e:
        mc.lod_encode_attr(False, lp, [lift_params([0]), lift_params([0])], offsets, xyz, attrs)
    assert "Morton order" in str(e.value)
    mc.close()

"""GPU parity of the tile kernels (raht_tile.hpp: sub-node prediction off) at
sizes that span many tiles, the coarse / per-level boundary, key windows that
a neighbour leaves (global fallback), tiles over many tiny slices, region QPs,
the non-extension mode -- and the int32 wrap of the attribute sums, which is
the reference's own arithmetic (tmc3/RAHT.cpp:196 accumulates in `int`).
Bit-exact against the compiled reference (or the C oracle where it did not
travel)."""
import numpy as np
import pytest

import oracle_loader as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


def checker():
    return ol.ref() if ol.ref_available() else ol.oracle()


def cloud(kind, n, seed):
    from mpeg_pcc_tmc13_amd import synth
    if kind == "dense":
        xyz, a = synth.dense_cloud(n, seed=seed, bits=9)
    elif kind == "lidar":
        xyz, a = synth.lidar_cloud(n, seed=seed)
    else:
        xyz, a = synth.random_cloud(n, seed=seed, bits=7, c=2, dup_fraction=0.1)
    morton, a, _ = synth.sort_by_morton(xyz, a)
    return morton, a


@pytest.mark.parametrize("kind,n", [("dense", 150_000), ("lidar", 180_000), ("random", 120_000)])
@pytest.mark.parametrize("search_range", [8, 2500, 50000])
def test_multi_tile_lossy_and_decode(kind, n, search_range, ctx):
    from mpeg_pcc_tmc13_amd import raht_params
    morton, attrs = cloud(kind, n, seed=11)
    c = attrs.shape[1]
    p = raht_params(qp=31, subnode=False, search_range=search_range)
    want_co, want_rec = checker().raht_forward(p, morton, attrs)
    co, rec = ctx.raht_forward(p, morton, attrs)
    np.testing.assert_array_equal(co, want_co)
    np.testing.assert_array_equal(rec, want_rec)
    np.testing.assert_array_equal(ctx.raht_inverse(p, morton, want_co, c), want_rec)


@pytest.mark.parametrize("kind,n", [("dense", 90_000), ("lidar", 110_000)])
@pytest.mark.parametrize("variant", ["haar", "noext", "nopred", "qp4", "thresholds"])
def test_multi_tile_variants(kind, n, variant, ctx):
    from mpeg_pcc_tmc13_amd import raht_params
    morton, attrs = cloud(kind, n, seed=12)
    c = attrs.shape[1]
    kw = dict(qp=28, subnode=False, search_range=2500 if kind == "lidar" else 50000)
    if variant == "haar":
        kw.update(qp=4, haar=True, chroma_offset=0)
    elif variant == "noext":
        kw.update(extension=False)
    elif variant == "nopred":
        kw.update(prediction=False)
    elif variant == "qp4":
        kw.update(qp=4)
    else:
        kw.update(threshold0=4, threshold1=9)
    p = raht_params(**kw)
    want_co, want_rec = checker().raht_forward(p, morton, attrs)
    co, rec = ctx.raht_forward(p, morton, attrs)
    np.testing.assert_array_equal(co, want_co)
    np.testing.assert_array_equal(rec, want_rec)
    np.testing.assert_array_equal(ctx.raht_inverse(p, morton, want_co, c), want_rec)


def test_multi_tile_region_qp_and_layers(ctx):
    from mpeg_pcc_tmc13_amd import raht_params
    morton, attrs = cloud("dense", 70_000, seed=13)
    n, c = attrs.shape
    rng = np.random.default_rng(3)
    qp_off = np.zeros((n, 2), np.int32)
    qp_off[n // 3: 2 * n // 3] = (-5, 2)
    qp_off[rng.integers(0, n, 500)] = (3, -1)
    p = raht_params(qp=30, subnode=False, search_range=50000)
    p.set_layers([(30, -1), (34, 0), (26, -2)])
    want_co, want_rec = checker().raht_forward(p, morton, attrs, qp_off)
    co, rec = ctx.raht_forward(p, morton, attrs, qp_off)
    np.testing.assert_array_equal(co, want_co)
    np.testing.assert_array_equal(rec, want_rec)
    np.testing.assert_array_equal(ctx.raht_inverse(p, morton, want_co, c, qp_off), want_rec)


@pytest.mark.parametrize("haar", [False, True])
def test_tiles_over_many_small_slices(haar, ctx):
    """Thousands of slices of 1..60 points plus a few large ones in one batch: a
    tile of 1024 parents spans far more than the 8 slices whose plan fits LDS,
    and whole slices are coarse."""
    import torch
    from mpeg_pcc_tmc13_amd import raht_params, synth
    rng = np.random.default_rng(17)
    sizes = [int(x) for x in rng.integers(1, 60, 1500)]
    sizes[100] = 30_000
    sizes[700] = 9_000
    sizes[1499] = 5_000
    p = (raht_params(qp=4, haar=True, chroma_offset=0, subnode=False) if haar
         else raht_params(qp=33, subnode=False))
    ms, as_ = [], []
    for i, n in enumerate(sizes):
        xyz, col = synth.random_cloud(n, seed=900 + i, bits=6 if n > 1000 else 3, c=3,
                                      dup_fraction=0.1 if n > 10 else 0.0)
        m, a, _ = synth.sort_by_morton(xyz, col)
        ms.append(m)
        as_.append(a)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    dev = torch.device("cuda:0")
    d_m = torch.from_numpy(np.concatenate(ms)).to(dev)
    d_a = torch.from_numpy(np.concatenate(as_).reshape(-1)).to(dev)
    d_c = torch.zeros(3 * int(offsets[-1]), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.set_morton_bits(18)
    ctx.dev_raht_forward(p, offsets, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 3)
    ctx.synchronize()
    rec, co = d_a.cpu().numpy(), d_c.cpu().numpy()
    d_a2 = torch.zeros_like(d_a)
    ctx.dev_raht_inverse(p, offsets, d_m.data_ptr(), d_a2.data_ptr(), d_c.data_ptr(), 3)
    ctx.synchronize()
    ctx.set_morton_bits(0)
    inv = d_a2.cpu().numpy()
    o = ol.oracle()
    for i, n in enumerate(sizes):
        o_co, o_rec = o.raht_forward(p, ms[i], as_[i])
        b = int(offsets[i])
        np.testing.assert_array_equal(co[3 * b:3 * (b + n)], o_co, err_msg=f"slice {i}")
        np.testing.assert_array_equal(rec[3 * b:3 * (b + n)].reshape(n, 3), o_rec, err_msg=f"slice {i}")
        np.testing.assert_array_equal(inv[3 * b:3 * (b + n)].reshape(n, 3), o_rec, err_msg=f"slice {i}")


@pytest.mark.parametrize("subnode", [False, True])
def test_attribute_sums_wrap_like_the_reference(subnode, ctx):
    """16-bit attributes whose node sums exceed 2^31: the reference accumulates
    them in `int` (tmc3/RAHT.cpp:131,196; attrsLf is std::vector<int>), the
    device takes int32 differences of a modular prefix sum -- the same value."""
    from mpeg_pcc_tmc13_amd import raht_params, synth
    xyz, _ = synth.dense_cloud(300_000, seed=3, bits=9)
    rng = np.random.default_rng(5)
    attrs = rng.integers(40000, 65536, size=(len(xyz), 1)).astype(np.int32)
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    assert int(attrs.sum()) > 2 ** 32
    p = raht_params(qp=40, subnode=subnode, bitdepth=16)
    want_co, want_rec = checker().raht_forward(p, morton, attrs)
    co, rec = ctx.raht_forward(p, morton, attrs)
    np.testing.assert_array_equal(co, want_co)
    np.testing.assert_array_equal(rec, want_rec)


def test_morton_bits_hint_too_small_is_reported(ctx):
    """The device tier sizes the top levels from the hint: codes wider than it
    raise an error at the next synchronisation instead of a wrong result."""
    import torch
    from mpeg_pcc_tmc13_amd import raht_params, synth
    from mpeg_pcc_tmc13_amd._lib import GpccError
    xyz, col = synth.dense_cloud(20_000, seed=2, bits=9)
    m, a, _ = synth.sort_by_morton(xyz, col)
    dev = torch.device("cuda:0")
    d_m, d_a = torch.from_numpy(m).to(dev), torch.from_numpy(a.reshape(-1)).to(dev)
    d_c = torch.zeros(3 * len(m), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    offsets = np.array([0, len(m)], np.int64)
    p = raht_params(qp=30, subnode=False)
    ctx.set_morton_bits(12)  # the codes have 27 bits
    ctx.dev_raht_forward(p, offsets, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 3)
    with pytest.raises(GpccError) as ei:
        ctx.synchronize()
    assert ei.value.code == -1 and "morton" in str(ei.value).lower()
    # the error is reported once; the context is usable afterwards
    ctx.set_morton_bits(27)
    d_a.copy_(torch.from_numpy(a.reshape(-1)).to(dev))
    torch.cuda.synchronize()
    ctx.dev_raht_forward(p, offsets, d_m.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 3)
    ctx.synchronize()
    ctx.set_morton_bits(0)
    o_co, o_rec = ol.oracle().raht_forward(p, m, a)
    np.testing.assert_array_equal(d_c.cpu().numpy(), o_co)


def test_failure_leaves_the_callers_attributes_alone(ctx):
    """Host tier: the source attributes are overwritten only when the call
    succeeds (a shim falls back to the reference with the same buffer)."""
    from mpeg_pcc_tmc13_amd import raht_params
    from mpeg_pcc_tmc13_amd._lib import GpccError
    morton = np.array([5, 3, 9], dtype=np.int64)  # unsorted: refused
    attrs = np.arange(9, dtype=np.int32).reshape(3, 3)
    keep = attrs.copy()
    with pytest.raises(GpccError):
        ctx.raht_forward(raht_params(subnode=False), morton, attrs)
    np.testing.assert_array_equal(attrs, keep)


def test_context_counters(ctx):
    from mpeg_pcc_tmc13_amd import raht_params
    from mpeg_pcc_tmc13_amd._lib import GpccError
    morton, attrs = cloud("dense", 5_000, seed=1)
    before = ctx.stats()
    ctx.raht_forward(raht_params(subnode=False), morton, attrs)
    with pytest.raises(GpccError):
        ctx.raht_forward(raht_params(subnode=False), morton[::-1].copy(), attrs)
    after = ctx.stats()
    assert after["calls_ok"] == before["calls_ok"] + 1
    assert after["calls_failed"] == before["calls_failed"] + 1
    assert after["points_ok"] == before["points_ok"] + len(morton)

"""Binarisation of the residual symbols (the non-arithmetic-coder part of
PCCResidualsEncoder, tmc3/AttributeEncoder.cpp:227-307).

CPU: the C oracle's decisions, fed to the reference's own arithmetic coder and
context models, give byte for byte what the reference class produces for the
same symbols -- for random symbol streams incl. long runs and large magnitudes,
and for the symbols of real slices.  GPU: the device's decisions equal the
oracle's, and the whole chain positions + attributes -> device transform ->
device zero-run formation -> device binarisation -> reference arithmetic
coder reproduces the reference operator's payload."""
import numpy as np
import pytest

import lod_helpers as lh
import oracle_loader as ol

needs_entropy = pytest.mark.skipif(not lh.entropy_available(), reason="libtmc3_entropy.so not built")


def random_symbols(rng, m, c, big=False):
    runs = rng.choice([0, 0, 0, 1, 2, 3, 5, 9, 10, 11, 12, 40, 1000, 70000], size=m).astype(np.int32)
    mag = rng.choice([1, 1, 1, 2, 3, 4, 7, 8, 15, 16, 100, 5000] + ([1 << 20, (1 << 30) - 1] if big else []), size=(m, c))
    vals = (mag * rng.choice([-1, 1], size=(m, c))).astype(np.int32)
    # a coded position has at least one non-zero component; others may be zero
    if c == 3:
        zero = rng.random((m, c)) < 0.4
        zero[np.arange(m), rng.integers(0, 3, m)] = False
        vals[zero] = 0
    return runs, vals


@needs_entropy
@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("seed", range(4))
def test_oracle_decisions_drive_the_reference_coder(c, seed):
    rng = np.random.default_rng(8000 + seed)
    m = int(rng.integers(1, 3000))
    runs, vals = random_symbols(rng, m, c, big=seed == 3)
    trailing = int(rng.choice([0, 1, 17, 100000]))
    n_points = int(runs.sum() + m + trailing)
    want = lh.ref_entropy_encode_symbols(c, n_points, runs, vals, trailing)
    bins = lh.oracle_binarise_symbols(runs, vals, trailing, c)
    assert bins.max() >> 1 <= 31
    assert lh.ref_entropy_encode_bins(bins, n_points) == want


@needs_entropy
def test_empty_and_run_only_streams():
    e = np.zeros(0, np.int32)
    assert len(lh.oracle_binarise_symbols(e, e, 0, 3)) == 0
    bins = lh.oracle_binarise_symbols(e, e, 123456, 1)
    assert lh.ref_entropy_encode_bins(bins, 123456) == lh.ref_entropy_encode_symbols(1, 123456, e, e, 123456)


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("seed", range(3))
def test_device_decisions_equal_oracle(c, seed, ctx):
    rng = np.random.default_rng(8100 + seed)
    m = int(rng.integers(1, 200_000))
    runs, vals = random_symbols(rng, m, c, big=seed == 2)
    trailing = int(rng.choice([0, 5, 300000]))
    np.testing.assert_array_equal(ctx.binarise_symbols(runs, vals, trailing, c),
                                  lh.oracle_binarise_symbols(runs, vals, trailing, c))
    e = np.zeros(0, np.int32)
    np.testing.assert_array_equal(ctx.binarise_symbols(e, e, 7, c), lh.oracle_binarise_symbols(e, e, 7, c))
    assert len(ctx.binarise_symbols(e, e, 0, c)) == 0


@pytest.mark.gpu
@needs_entropy
@pytest.mark.parametrize("kind,c", [("dense", 3), ("lidar", 1)])
def test_slice_to_bitstream_through_device_binarisation(kind, c, ctx):
    """positions + attributes -> gpcc_raht_encode_attr_packed (device transform +
    zero runs) -> gpcc_binarise_symbols -> the reference's arithmetic coder ==
    the arithmetic-coded part of the payload AttributeEncoder::encode writes."""
    from mpeg_pcc_tmc13_amd import lod_params, raht_params, synth
    n = 120_000
    xyz, attrs = synth.dense_cloud(n, seed=9, bits=9) if kind == "dense" else synth.lidar_cloud(n, seed=9)
    n = len(xyz)
    rp = raht_params(qp=34, chroma_offset=-1 if c == 3 else 0, search_range=50000 if kind == "dense" else 2500)
    payload, rec_enc, _ = lh.ref_operator_roundtrip(lod_params(), 0, rp, 34, -1 if c == 3 else 0, 8, 1, xyz, attrs)
    want = payload[lh.ref_last_abh_size():]
    runs, vals, trailing, rec = ctx.raht_encode_attr_packed(rp, xyz, attrs, 8)
    np.testing.assert_array_equal(rec, rec_enc)
    bins = ctx.binarise_symbols(runs, vals, trailing, c)
    assert lh.ref_entropy_encode_bins(bins, n) == want

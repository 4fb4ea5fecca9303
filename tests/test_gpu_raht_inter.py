"""GPU: RAHT with attribute inter prediction on the device (gpcc_raht_forward_inter / _inverse_inter:
csrc/raht_inter.hpp, raht_tile.hpp with INTER) against the oracle and, where it is built, the compiled
reference: coefficients, encoder reconstruction, decoder output, per-layer modes, filter taps -- bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_loader as ol
from test_oracle_raht_inter import clouds, frame_of, run

pytestmark = pytest.mark.gpu

VARIANTS = [dict(subnode=False), dict(prediction=False), dict(subnode=False, qp=22), dict(subnode=False, extension=False),
            dict(subnode=False, qp=46, chroma_offset=0)]


def inter_params(depth, rdo, fest, skip):
    from mpeg_pcc_tmc13_amd import RahtInterParams
    return RahtInterParams(depth, rdo, fest, skip)


def check(ctx, p, morton, attrs, mref, aref, depth, rdo, fest, skip, tag, also_ref=False):
    rc, co_o, rec_o, modes_o, taps_o = run(ol.oracle().lib, "oracle_raht_inter", p, True, morton, attrs, None, mref, aref, depth, rdo, fest, skip)
    assert rc == 0
    ip = inter_params(depth, rdo, fest, skip)
    co, rec, modes, taps = ctx.raht_forward_inter(p, ip, morton, attrs, mref, aref)
    np.testing.assert_array_equal(taps, taps_o, err_msg=f"{tag} filter taps")
    np.testing.assert_array_equal(modes, modes_o, err_msg=f"{tag} layer modes")
    np.testing.assert_array_equal(co, co_o, err_msg=f"{tag} coefficients")
    np.testing.assert_array_equal(rec, rec_o, err_msg=f"{tag} encoder reconstruction")
    dec = ctx.raht_inverse_inter(p, ip, morton, co_o, attrs.shape[1], mref, aref, modes_o, taps_o)
    np.testing.assert_array_equal(dec, rec_o, err_msg=f"{tag} decoder")
    if also_ref and ol.ref_available():
        rc, co_r, rec_r, modes_r, taps_r = run(ol.ref().lib, "ref_raht_inter", p, True, morton, attrs, None, mref, aref, depth, rdo, fest, skip)
        assert rc == 0
        np.testing.assert_array_equal(co, co_r, err_msg=f"{tag} coefficients vs the compiled reference")
        np.testing.assert_array_equal(modes, modes_r)
        np.testing.assert_array_equal(taps, taps_r)
    return modes_o, taps_o


@pytest.mark.parametrize("vi", range(len(VARIANTS)))
@pytest.mark.parametrize("rdo,fest", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_inter_raht_against_the_oracle(vi, rdo, fest):
    from mpeg_pcc_tmc13_amd import context, raht_params, synth
    ctx = context(0)
    kw = VARIANTS[vi]
    rng = np.random.default_rng(3)
    seen_modes, seen_taps = set(), set()
    for name, xyz, attrs in clouds():
        if name == "one":
            continue
        morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
        for shift, jitter in ((0, 2), (0, 40), (40, 6)):
            mref, aref = frame_of(xyz, attrs, rng, shift=shift, jitter=jitter)
            for depth, skip in ((15, 0), (2, 3), (15, 3)):
                m, t = check(ctx, raht_params(**kw), morton, a_sorted, mref, aref, depth, rdo, fest, skip,
                             f"{name} {kw} shift{shift} jitter{jitter} depth{depth} skip{skip} rdo{rdo} fest{fest}",
                             also_ref=(depth == 15 and skip == 0))
                seen_modes.update(m.tolist())
                seen_taps.update(t.tolist())
    if rdo and kw.get("prediction", True):
        assert seen_modes == {0, 1}, seen_modes
    if fest:
        assert len(seen_taps) > 1, seen_taps


@pytest.mark.parametrize("kind,n,c", [("dense", 200000, 3), ("lidar", 300000, 1)])
def test_inter_raht_large_frames(kind, n, c):
    """frames of a size where levels span many tiles and rate chains many chunks; the reference's default tools"""
    from mpeg_pcc_tmc13_amd import context, raht_params, synth
    ctx = context(0)
    rng = np.random.default_rng(11)
    if kind == "dense":
        xyz, attrs = synth.dense_cloud(n, seed=5, bits=10)
    else:
        xyz, attrs = synth.lidar_cloud(n, seed=6)
        attrs = attrs >> 8 if attrs.max() > 255 else attrs
    attrs = attrs[:, :c]
    morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
    mref, aref = frame_of(xyz, attrs, rng, jitter=4)
    for rdo, fest in ((1, 0), (1, 1)):
        check(ctx, raht_params(subnode=False), morton, a_sorted, mref, aref, 15, rdo, fest, 3, f"{kind} rdo{rdo} fest{fest}")


def test_declined_configurations():
    from mpeg_pcc_tmc13_amd import context, raht_params, synth
    from mpeg_pcc_tmc13_amd._lib import GpccError
    ctx = context(0)
    xyz, attrs = synth.dense_cloud(2000, seed=3, bits=6)
    morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
    for kw in (dict(), dict(haar=True, qp=4, chroma_offset=0, subnode=False)):
        with pytest.raises(GpccError) as e:
            ctx.raht_forward_inter(raht_params(**kw), inter_params(15, 1, 0, 0), morton, a_sorted, morton, a_sorted)
        assert e.value.code == -2, e.value

"""GPU: RAHT with attribute inter prediction on the device (gpcc_raht_forward_inter / _inverse_inter:
csrc/raht_inter.hpp, raht_tile.hpp with INTER) against the oracle and, where it is built, the compiled
reference: coefficients, encoder reconstruction, decoder output, per-layer modes, filter taps -- bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_loader as ol
from test_oracle_raht_inter import clouds, frame_of, region_offsets, run, run_qp

pytestmark = pytest.mark.gpu

VARIANTS = [dict(subnode=False), dict(prediction=False), dict(subnode=False, qp=22), dict(subnode=False, extension=False),
            dict(subnode=False, qp=46, chroma_offset=0),
            # the reference's default flag: sub-node prediction
            dict(), dict(extension=False), dict(qp=22), dict(qp=46, chroma_offset=0),
            # the integer Haar kernel (the lossless configurations), without and with sub-node prediction
            dict(haar=True, qp=4, chroma_offset=0, subnode=False), dict(haar=True, qp=4, chroma_offset=0)]


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
            if kw.get("haar") and len(mref) > 1 and (int(mref[0] ^ mref[-1]).bit_length() - int(morton[0] ^ morton[-1]).bit_length()) % 3:
                continue  # declined: the two trees do not line up on octree levels (test_declined_configurations)
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


@pytest.mark.parametrize("kw", [dict(subnode=False), dict(), dict(haar=True, qp=4, chroma_offset=0), dict(subnode=False, extension=False),
                                dict(qp=22)])
def test_inter_raht_with_region_qp_offsets(kw):
    """per-point QP offsets of a region (QpSet::regionQpOffset) together with inter prediction, against the oracle and
    the compiled reference: every kernel family (tile, dependency, integer Haar)"""
    from mpeg_pcc_tmc13_amd import context, raht_params, synth
    ctx = context(0)
    rng = np.random.default_rng(8)
    for name, xyz, attrs in clouds():
        if name in ("one", "tiny"):
            continue
        morton, a_sorted, order = synth.sort_by_morton(xyz, attrs)
        q = region_offsets(xyz[order], rng)
        mref, aref = frame_of(xyz, attrs, rng, jitter=4)
        if kw.get("haar") and (int(mref[0] ^ mref[-1]).bit_length() - int(morton[0] ^ morton[-1]).bit_length()) % 3:
            continue
        for rdo, fest in ((1, 1), (1, 0), (0, 0)):
            p = raht_params(**kw)
            ip = inter_params(15, rdo, fest, 3)
            rc, co_o, rec_o, modes_o, taps_o = run_qp(ol.oracle().lib, "oracle_raht_inter_qp", p, True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3, q)
            assert rc == 0
            co, rec, modes, taps = ctx.raht_forward_inter(p, ip, morton, a_sorted, mref, aref, qp_off=q)
            np.testing.assert_array_equal(taps, taps_o)
            np.testing.assert_array_equal(modes, modes_o)
            np.testing.assert_array_equal(co, co_o)
            np.testing.assert_array_equal(rec, rec_o)
            dec = ctx.raht_inverse_inter(p, ip, morton, co_o, a_sorted.shape[1], mref, aref, modes_o, taps_o, qp_off=q)
            np.testing.assert_array_equal(dec, rec_o)
            if ol.ref_available():
                rc, co_r, _, modes_r, taps_r = run_qp(ol.ref().lib, "ref_raht_inter_qp", p, True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3, q)
                assert rc == 0
                np.testing.assert_array_equal(co, co_r)
                np.testing.assert_array_equal(modes, modes_r)
                np.testing.assert_array_equal(taps, taps_r)


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
    for kw in (dict(subnode=False), dict()):
        for rdo, fest in ((1, 0), (1, 1)):
            check(ctx, raht_params(**kw), morton, a_sorted, mref, aref, 15, rdo, fest, 3, f"{kind} {kw} rdo{rdo} fest{fest}")


def test_declined_configurations():
    from mpeg_pcc_tmc13_amd import context, raht_params, synth
    from mpeg_pcc_tmc13_amd._lib import GpccError
    ctx = context(0)
    xyz, attrs = synth.dense_cloud(2000, seed=3, bits=6)
    morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
    # the integer Haar kernel with a frame whose tree does not line up on octree levels
    haar = raht_params(haar=True, qp=4, chroma_offset=0)
    taller = morton.copy()
    taller[-1] |= 1 << (int(morton[0] ^ morton[-1]).bit_length() + 1)
    with pytest.raises(GpccError) as e:
        ctx.raht_forward_inter(haar, inter_params(15, 1, 0, 0), morton, a_sorted, taller, a_sorted)
    assert e.value.code == -2, e.value



# ---- seam 1 with inter slices: libtmc3_shim.so in a process of its own (tests/raht_inter_shim_worker.py), the
#      unmodified libtmc3_ref.so here ---------------------------------------------------------------------------
def _shim_worker(what, tmp_path):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libtmc3_shim.so")):
        pytest.skip("libtmc3_shim.so absent")
    out = str(tmp_path / "shim.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "raht_inter_shim_worker.py"), what, out],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out, allow_pickle=True), r.stderr


def seam1_cases():
    from mpeg_pcc_tmc13_amd import raht_params, synth
    rng = np.random.default_rng(5)
    xyz, attrs = synth.dense_cloud(30000, seed=9, bits=8)
    morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
    # the frame inside the current frame's bounding cube and with its corners, so that the two trees have the same
    # height (the integer Haar kernel is declined otherwise)
    keep = rng.random(len(xyz)) > 0.1
    keep[np.argmin(xyz.sum(1))] = keep[np.argmax(xyz.sum(1))] = True
    xr = np.clip(xyz + rng.integers(-1, 2, size=xyz.shape), xyz.min(0), xyz.max(0))[keep].astype(np.int32)
    xr[0], xr[-1] = xyz[np.argmin(xyz.sum(1))], xyz[np.argmax(xyz.sum(1))]
    ar = np.clip(attrs + rng.integers(-4, 5, size=attrs.shape), 0, 255)[keep].astype(np.int32)
    mref, aref = synth.sort_by_morton(xr, ar)[:2]
    assert int(mref[0] ^ mref[-1]).bit_length() == int(morton[0] ^ morton[-1]).bit_length()
    # (parameters, layer decision, estimated taps, runs on the device)
    cases = [(dict(subnode=False), 1, 1, True), (dict(subnode=False), 1, 0, True), (dict(prediction=False), 0, 0, True),
             (dict(), 1, 0, True), (dict(haar=True, qp=4, chroma_offset=0), 1, 0, True),
             (dict(haar=True, qp=4, chroma_offset=0, subnode=False), 1, 1, True)]
    return morton, a_sorted, mref, aref, [(raht_params(**kw), rdo, fest, dev) for kw, rdo, fest, dev in cases]


def test_seam1_runs_inter_slices_on_the_device(tmp_path):
    """the reference's own callers' entry points (pcc::regionAdaptiveHierarchicalTransform / ...Inverse..., replaced
    by shim/RAHT_mi355.cpp) with attrInterPredParams.enableAttrInterPred: the device runs the slice, the modes and
    taps land in the reference's vectors, everything equals the unmodified library's (the reference's default flags
    included: sub-node prediction + per-layer decision, and the integer Haar kernel)"""
    if not ol.ref_available():
        pytest.skip("compiled reference absent")
    got, log = _shim_worker("function", tmp_path)
    morton, a_sorted, mref, aref, cases = seam1_cases()
    for i, (p, rdo, fest, dev) in enumerate(cases):
        rc, co_r, rec_r, modes_r, taps_r = run(ol.ref().lib, "ref_raht_inter", p, True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3)
        assert rc == 0
        np.testing.assert_array_equal(got[f"co{i}"], co_r)
        np.testing.assert_array_equal(got[f"rec{i}"], rec_r)
        np.testing.assert_array_equal(got[f"modes{i}"], modes_r)
        np.testing.assert_array_equal(got[f"taps{i}"], taps_r)
        np.testing.assert_array_equal(got[f"dec{i}"], rec_r)
        assert tuple(got[f"calls{i}"]) == ((2, 0) if dev else (0, 2)), (i, got[f"calls{i}"], log)


# (parameters, per-layer decision, estimated taps, QP region); the second one is the reference's default configuration,
# the last carries a QP region in the attribute brick header (origin, size, (luma, chroma) offsets)
OPERATOR_REGION = ((110000, 105000, 129000), (30000, 30000, 3000), (5, 0))  # (inside the lidar cloud of operator_case)
OPERATOR_CASES = [(dict(subnode=False), 1, 1, None), (dict(), 1, 0, None), (dict(subnode=False), 0, 0, None),
                  (dict(), 1, 0, OPERATOR_REGION)]


def operator_case():
    from mpeg_pcc_tmc13_amd import synth
    rng = np.random.default_rng(11)
    xyz, attrs = synth.lidar_cloud(60000, seed=61)
    attrs = attrs[:, :1].copy()
    if attrs.max() > 255:
        attrs = attrs >> 8
    keep = rng.random(len(xyz)) > 0.1
    xr = np.clip(xyz + rng.integers(-1, 2, size=xyz.shape), 0, None)[keep].astype(np.int32)
    ar = np.clip(attrs + rng.integers(-6, 7, size=attrs.shape), 0, 255)[keep].astype(np.int32)
    return xyz, attrs, xr, ar


def test_operator_with_inter_raht_on_the_device(tmp_path):
    """AttributeEncoder::encode (encodeReflectancesTransformRaht with a reference frame) + AttributeDecoder::decode of
    the library with the link seams replaced: the attribute brick's payload, the reconstructions, the signalled modes
    and taps equal the unmodified operator's, and the RAHT calls ran on the device"""
    from test_oracle_raht_inter import _operator_roundtrip
    from mpeg_pcc_tmc13_amd import raht_params
    if not ol.ref_available():
        pytest.skip("compiled reference absent")
    got, log = _shim_worker("operator", tmp_path)
    xyz, attrs, xr, ar = operator_case()
    import lod_helpers as lh
    plain = None
    for i, (kw, rdo, fest, region) in enumerate(OPERATOR_CASES):
        lh.ref_set_qp_region(region)
        try:
            want = _operator_roundtrip(raht_params(**kw), 34, xyz, attrs, xr, ar, 15, rdo, fest, 3)
        finally:
            lh.ref_set_qp_region(None)
        if region is None and (kw, rdo, fest) == (dict(), 1, 0):
            plain = want[0]
        if region is not None:
            assert want[0] != plain, "the region changes nothing: not a test of it"
        assert got[f"payload{i}"].tobytes() == want[0], "payload"
        for j, name in enumerate(("enc", "dec", "modes", "taps")):
            np.testing.assert_array_equal(got[f"{name}{i}"], want[1 + j])
        calls = tuple(got[f"calls{i}"])
        assert calls[0] >= 2 and calls[1] == 0, (calls, log)

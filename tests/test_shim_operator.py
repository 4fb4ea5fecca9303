"""The drop-in boundary end to end at the OPERATOR: the reference's
AttributeEncoder::encode + AttributeDecoder::decode (its entropy coder, its
HLS, its slice drivers -- the objects of the unmodified build) linked with the
two replacement translation units instead of RAHT.o / AttributeCommon.o, so
the RAHT transform and the LoD build of every slice run on the MI355X
(oracle/_ref/libtmc3_shim.so, oracle/Makefile).  The payload it writes -- the
attribute brick of the bitstream -- must be byte-identical to the unmodified
build's (the criterion of the reference's own scripts/Makefile.tmc13-step:
md5 of the coded stream, decoder output == encoder reconstruction), and the
shims' counters must say the device did the work: a silent CPU fallback would
also be "identical"."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_loader as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "_ref", "libtmc3_shim.so")
needs = pytest.mark.skipif(not (os.path.exists(SHIM) and ol.ref_available()), reason="libtmc3_shim.so / libtmc3_ref.so not built")

# BASELINE configs[0/1] shape: octree-raht lossless-geom lossy-attrs (r04: qp 34,
# qpChromaOffset -1, reference default flags), colour and reflectance; the
# lossless RAHT configuration; a lifting configuration (seam 2)
CASES = {
    "raht_colour_100k": dict(cloud="dense", n=100_000, seed=4, transform=0, qp=34, chroma=-1, subnode=1, search_range=50000),
    "raht_refl_lidar_100k": dict(cloud="lidar", n=100_000, seed=5, transform=0, qp=34, chroma=0, subnode=1, search_range=2500),
    "raht_colour_sub0": dict(cloud="dense", n=60_000, seed=6, transform=0, qp=40, chroma=-1, subnode=0, search_range=50000),
    "raht_haar_lossless": dict(cloud="dense", n=50_000, seed=7, transform=0, qp=4, chroma=0, subnode=1, haar=1, search_range=50000),
    "lifting_colour_100k": dict(cloud="dense", n=100_000, seed=8, transform=2, qp=34, chroma=-1, subnode=1, search_range=50000),
    "lifting_scalable_colour_60k": dict(cloud="dense", n=60_000, seed=10, transform=2, qp=34, chroma=-1, subnode=1, search_range=50000,
                                        scalable=1),
}


def run_worker(case, strict):
    env = dict(os.environ)
    if strict:
        env["GPCC_STRICT"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_operator_worker.py"), json.dumps(case)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr


def unmodified(case):
    """payload md5 / length / reconstruction digest of the unmodified build for the case"""
    import lod_helpers as lh
    lh.ref_set_qp_region(case.get("region"))
    try:
        return _unmodified(case)
    finally:
        lh.ref_set_qp_region(None)  # (the harness keeps the region for every header it builds: other tests share it)


def _unmodified(case):
    import lod_helpers as lh
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shim_operator_worker as w
    if case.get("multi_slice"):
        offs, xyz, col, refl, lpa, ta, lpb, tb = w.multi_slice_case(case)
        payload, lens, enc2, dec2, reused = lh.ref_multi_slice_roundtrip(lpa, ta, lpb, tb, case["qp"], offs, xyz, col, refl)
        rec_enc = np.concatenate([enc2[0].reshape(-1), enc2[1]])
        rec_dec = np.concatenate([dec2[0].reshape(-1), dec2[1]])
        np.testing.assert_array_equal(rec_enc, rec_dec)
        assert reused == [1] * (len(offs) - 1)
        return hashlib.md5(payload).hexdigest(), len(payload), w.digest(rec_enc)
    if case.get("two_attr"):
        xyz, col, refl, lpa, ta, lpb, tb = w.two_attr_case(case)
        payload, enc2, dec2, reused = lh.ref_two_attr_roundtrip(lpa, ta, lpb, tb, case["qp"], xyz, col, refl)
        rec_enc = np.concatenate([enc2[0].reshape(-1), enc2[1]])
        rec_dec = np.concatenate([dec2[0].reshape(-1), dec2[1]])
        np.testing.assert_array_equal(rec_enc, rec_dec)
        assert reused == tuple(case.get("expect_reused", (1, 1)))
        return hashlib.md5(payload).hexdigest(), len(payload), w.digest(rec_enc)
    if case.get("inter"):
        xyz, attrs, xr, ar, lp = w.inter_case(case)
        payload, rec_enc, rec_dec = lh.ref_inter_roundtrip(lp, case["transform"], case["qp"], 8, case.get("direct", 3), xyz, attrs,
                                                           xr, ar, case.get("search_range", 128), 1, threshold=4)
    elif case["transform"] == 1:
        xyz, attrs, lp, pp, thr, qp = w.pred_case(case)
        payload, rec_enc, rec_dec, _ = lh.ref_pred_roundtrip(lp, pp, thr, qp, 0, xyz, attrs)
    else:
        xyz, attrs, rp, lp = w.make_case(case)
        payload, rec_enc, rec_dec = lh.ref_operator_roundtrip(
            lp, case["transform"], rp, case["qp"], case["chroma"], 8 if attrs.shape[1] == 3 else case.get("bitdepth", 8),
            1, xyz, attrs)
    np.testing.assert_array_equal(rec_enc, rec_dec)  # the reference's own conformance criterion
    return hashlib.md5(payload).hexdigest(), len(payload), w.digest(rec_enc)


# attribute inter prediction: the LoD structure of such a slice comes from
# gpcc_lod_build_inter through AttributeLods::generate; the transform over it is the reference's
INTER_CASES = {
    "inter_lifting_refl_lidar_40k": dict(inter=1, cloud="lidar", n=40_000, seed=21, transform=2, qp=28),
    "inter_lifting_refl_dense_30k": dict(inter=1, cloud="dense", n=30_000, seed=22, transform=2, qp=10, search_range=64),
    "inter_pred_refl_lidar_30k": dict(inter=1, cloud="lidar", n=30_000, seed=23, transform=1, qp=16, direct=3),
}


@needs
def test_operator_with_shims_inter_slice_falls_back_without_gpu():
    from mpeg_pcc_tmc13_amd import _lib
    if _lib.load().gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    case = dict(INTER_CASES["inter_lifting_refl_lidar_40k"], n=5000)
    got, err = run_worker(case, strict=False)
    md5, ln, rec = unmodified(case)
    assert (got["payload_md5"], got["payload_len"], got["rec_enc_md5"], got["rec_dec_md5"]) == (md5, ln, rec, rec)
    assert got["lod_device"] == 0 and got["lod_cpu"] == 2


@needs
def test_operator_with_shims_falls_back_without_gpu():
    """CPU box: the same library still produces the reference's bitstream, every
    call counted as a fallback."""
    from mpeg_pcc_tmc13_amd import _lib
    if _lib.load().gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    case = dict(CASES["raht_colour_sub0"], n=8000)
    got, err = run_worker(case, strict=False)
    md5, ln, rec = unmodified(case)
    assert (got["payload_md5"], got["payload_len"], got["rec_enc_md5"], got["rec_dec_md5"]) == (md5, ln, rec, rec)
    assert got["raht_device"] == 0 and got["raht_cpu"] == 2


@needs
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_operator_bitstream_identical_with_device_inside(name):
    case = CASES[name]
    got, err = run_worker(case, strict=True)  # GPCC_STRICT=1: a fallback aborts the worker
    md5, ln, rec = unmodified(case)
    assert got["payload_len"] == ln and got["payload_md5"] == md5, "attribute payload differs from the unmodified build"
    assert got["rec_enc_md5"] == rec and got["rec_dec_md5"] == rec
    assert "falls back" not in err
    if case["transform"] == 0:
        # one forward (encoder) + one inverse (decoder) transform of the slice
        assert (got["raht_device"], got["raht_cpu"]) == (2, 0)
    else:
        # AttributeLods::generate once in the encoder, once in the decoder
        assert (got["lod_device"], got["lod_cpu"]) == (2, 0)


# ---- seam 3: the operator factories (libtmc3_shim3.so) --------------------------------------
SHIM3 = os.path.join(ROOT, "oracle", "_ref", "libtmc3_shim3.so")
needs3 = pytest.mark.skipif(not (os.path.exists(SHIM3) and ol.ref_available()), reason="libtmc3_shim3.so / libtmc3_ref.so not built")

# lifting (cfg/octree-liftt-ctc-*: colour with last-component prediction, reflectance) and the
# predicting transform (cfg/octree-predt-ctc-*: three direct predictors, inter-component
# prediction; a LiDAR reflectance slice; no direct predictors with neighbour-weighted quantisation)
CASES3 = {
    "lifting_colour_100k": dict(cloud="dense", n=100_000, seed=8, transform=2, qp=34, chroma=-1, subnode=1, search_range=50000),
    "lifting_refl_lidar_60k": dict(cloud="lidar", n=60_000, seed=9, transform=2, qp=28, chroma=0, subnode=1, search_range=2500),
    "lifting_scalable_colour_60k": dict(cloud="dense", n=60_000, seed=10, transform=2, qp=34, chroma=-1, subnode=1, search_range=50000,
                                        scalable=1),
    "lifting_scalable_refl_lidar_40k": dict(cloud="lidar", n=40_000, seed=11, transform=2, qp=28, chroma=0, subnode=1, search_range=2500,
                                            scalable=1, neigh_range=20),
    "pred_dense_ctc": dict(transform=1, pred_case="dense_ctc"),
    "pred_lidar_refl_ctc": dict(transform=1, pred_case="lidar_refl_ctc"),
    "pred_dense_nodirect_qnw": dict(transform=1, pred_case="dense_nodirect_qnw"),
    "pred_dense_scalable": dict(transform=1, pred_case="dense_scalable"),
    # RAHT slices (round 4): the whole slice driver on the device -- Morton sort, transform, zero runs,
    # binarisation -- colour and reflectance, default flags / sub-node prediction off / integer Haar
    "raht_colour_sub0": dict(cloud="dense", n=60_000, seed=6, transform=0, qp=40, chroma=-1, subnode=0, search_range=50000),
    "raht_colour_100k": dict(cloud="dense", n=100_000, seed=4, transform=0, qp=34, chroma=-1, subnode=1, search_range=50000),
    "raht_refl_lidar_100k": dict(cloud="lidar", n=100_000, seed=5, transform=0, qp=34, chroma=0, subnode=1, search_range=2500),
    "raht_haar_lossless": dict(cloud="dense", n=50_000, seed=7, transform=0, qp=4, chroma=0, subnode=1, haar=1, search_range=50000),
    "raht_refl_lidar_qp22": dict(cloud="lidar", n=80_000, seed=12, transform=0, qp=22, chroma=0, subnode=1, search_range=2500),
}


@needs3
def test_operator_factories_fall_back_without_gpu():
    """CPU box: the factories' encoder / decoder hand the slice to the reference's own."""
    from mpeg_pcc_tmc13_amd import _lib
    if _lib.load().gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    case = dict(CASES3["lifting_colour_100k"], n=6000, lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=False)
    md5, ln, rec = unmodified(case)
    assert (got["payload_md5"], got["payload_len"], got["rec_enc_md5"], got["rec_dec_md5"]) == (md5, ln, rec, rec)
    assert (got["enc_device"], got["enc_cpu"], got["dec_device"], got["dec_cpu"]) == (0, 1, 0, 1)


@needs3
def test_operator_factories_inter_slice_falls_back_without_gpu():
    from mpeg_pcc_tmc13_amd import _lib
    if _lib.load().gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    case = dict(INTER_CASES["inter_pred_refl_lidar_30k"], n=4000, lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=False)
    md5, ln, rec = unmodified(case)
    assert (got["payload_md5"], got["payload_len"], got["rec_enc_md5"], got["rec_dec_md5"]) == (md5, ln, rec, rec)
    assert (got["enc_device"], got["enc_cpu"], got["dec_device"], got["dec_cpu"]) == (0, 1, 0, 1)


@needs3
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES3))
def test_operator_bitstream_identical_with_device_coders_inside(name):
    """makeAttributeEncoder() / makeAttributeDecoder() of the drop-in build: transform, zero
    runs and binarisation of a lifting / predicting / RAHT slice on the MI355X, the decisions on
    the reference's arithmetic coder -- payload and reconstructions byte-identical to the unmodified
    build, the device counted once per direction, no fallback (GPCC_STRICT=1)."""
    case = dict(CASES3[name], lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=True)
    md5, ln, rec = unmodified(case)
    assert got["payload_len"] == ln and got["payload_md5"] == md5, "attribute payload differs from the unmodified build"
    assert got["rec_enc_md5"] == rec and got["rec_dec_md5"] == rec
    assert "falls back" not in err
    assert (got["enc_device"], got["enc_cpu"], got["dec_device"], got["dec_cpu"]) == (1, 0, 1, 0)
    # everything happens inside the one-call entries: neither AttributeLods::generate (seam 2) nor the
    # reference's RAHT slice drivers (and with them seam 1) are reached
    assert (got["lod_device"], got["lod_cpu"]) == (0, 0)
    assert (got["raht_device"], got["raht_cpu"]) == (0, 0)


# ---- the coder object's cached LoD structure: the second attribute of a slice runs over the
#      structure built for the FIRST whenever AttributeLods::isReusable passes -- which does not look at
#      attr_encoding (weight blending, the search inside a LoD: predicting transform only) or at
#      predictionWithDistributionEnabled ---------------------------------------------------------------
TWO_ATTR = {
    # colour: predicting transform with blended weights; reflectance: lifting over THOSE weights
    "pred_then_lifting": dict(two_attr=1, n=40_000, seed=31, qp=28, transforms=(1, 2)),
    # the other way round: the predicting transform runs over a lifting structure (weights not blended)
    "lifting_then_pred": dict(two_attr=1, n=40_000, seed=32, qp=28, transforms=(2, 1)),
    # both lifting; B asks for the distribution-aware third neighbour, A's structure has none
    "distribution_flag_differs": dict(two_attr=1, n=40_000, seed=33, qp=34, transforms=(2, 2),
                                      lod_a=dict(prediction_with_distribution_enabled=0),
                                      lod_b=dict(prediction_with_distribution_enabled=1)),
    # a field isReusable does compare: the objects are replaced, every attribute gets its own structure
    "neighbour_count_differs": dict(two_attr=1, n=30_000, seed=34, qp=34, transforms=(2, 2),
                                    lod_b=dict(num_pred_nearest_neighbours_minus1=1), expect_reused=(0, 0)),
}


@needs3
def test_two_attribute_cases_differ_from_fresh_structures():
    """the cases are not vacuous: coding B over its own fresh structure gives another payload"""
    import lod_helpers as lh
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shim_operator_worker as w
    case = dict(TWO_ATTR["pred_then_lifting"], n=6000)
    xyz, col, refl, lpa, ta, lpb, tb = w.two_attr_case(case)
    both, _, _, reused = lh.ref_two_attr_roundtrip(lpa, ta, lpb, tb, case["qp"], xyz, col, refl)
    assert reused == (1, 1)
    alone, _, _ = lh.ref_operator_roundtrip(lpb, tb, w.make_case(dict(cloud="dense", n=10, seed=1, qp=28, chroma=0, subnode=1,
                                                                       search_range=8, transform=2))[2],
                                            case["qp"], 0, 8, 0, xyz, refl.reshape(-1, 1))
    assert not both.endswith(alone)


@needs3
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TWO_ATTR))
def test_second_attribute_runs_over_the_first_attributes_structure(name):
    case = dict(TWO_ATTR[name], lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=True)
    md5, ln, rec = unmodified(case)
    assert got["payload_len"] == ln and got["payload_md5"] == md5, "payloads differ from the unmodified build"
    assert got["rec_enc_md5"] == rec and got["rec_dec_md5"] == rec
    assert got["reused"] == list(case.get("expect_reused", (1, 1)))
    assert (got["enc_device"], got["enc_cpu"], got["dec_device"], got["dec_cpu"]) == (2, 0, 2, 0)


# ---- several slices of a frame (tmc3/encoder.cpp:1044, 1209-1240, 1366-1368): new coder objects per slice, two
#      attributes inside a slice over one cached LoD structure, and the arithmetic coder's context memory of each
#      attribute CARRIED from slice to slice (entropy continuation) -- slices of different size ------------------------
MULTI_SLICE = {
    "three_slices_pred_then_lifting": dict(multi_slice=1, sizes=(30_000, 9_000, 17_000), seed=51, qp=28, transforms=(1, 2)),
    "four_slices_lifting_then_pred": dict(multi_slice=1, sizes=(12_000, 25_000, 3_000, 8_000), seed=55, qp=34, transforms=(2, 1)),
}


@needs3
def test_context_memory_is_carried_between_slices():
    """the case is not vacuous: the payload of a later slice coded on its own (fresh contexts) is another one"""
    import lod_helpers as lh
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shim_operator_worker as w
    case = dict(MULTI_SLICE["three_slices_pred_then_lifting"], sizes=(5000, 3000, 4000))
    offs, xyz, col, refl, lpa, ta, lpb, tb = w.multi_slice_case(case)
    payload, lens, _, _, reused = lh.ref_multi_slice_roundtrip(lpa, ta, lpb, tb, case["qp"], offs, xyz, col, refl)
    assert reused == [1, 1, 1] and len(lens) == 6 and sum(lens) == len(payload)
    a, b = int(offs[1]), int(offs[2])
    alone, _, _, _ = lh.ref_two_attr_roundtrip(lpa, ta, lpb, tb, case["qp"], xyz[a:b], col[a:b], refl[a:b])
    second = payload[lens[0] + lens[1]:lens[0] + lens[1] + lens[2] + lens[3]]
    assert second != alone


@needs3
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MULTI_SLICE))
def test_slices_of_a_frame_with_entropy_continuation(name):
    case = dict(MULTI_SLICE[name], lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=True)
    md5, ln, rec = unmodified(case)
    assert got["payload_len"] == ln and got["payload_md5"] == md5, "payloads differ from the unmodified build"
    assert got["rec_enc_md5"] == rec and got["rec_dec_md5"] == rec
    k = len(case["sizes"])
    assert got["reused"] == [1] * k
    assert (got["enc_device"], got["enc_cpu"], got["dec_device"], got["dec_cpu"]) == (2 * k, 0, 2 * k, 0)


@needs3
def test_two_attributes_on_a_cpu_box():
    """without a GPU the factories hand both attributes to the reference's objects: same bitstream"""
    from mpeg_pcc_tmc13_amd import _lib
    if _lib.load().gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    case = dict(TWO_ATTR["pred_then_lifting"], n=5000, lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=False)
    md5, ln, rec = unmodified(case)
    assert (got["payload_md5"], got["payload_len"], got["rec_enc_md5"], got["rec_dec_md5"]) == (md5, ln, rec, rec)
    assert got["reused"] == [1, 1]


# ---- QP regions (attr_region_* of the slice header): the one-call entries derive every point's offset
#      from its position on the device (round 4); seam 3 keeps such slices ---------------------------
REGION = ((40, 60, 30), (260, 200, 310), (-6, 3))   # origin, size, qp offsets (luma, chroma)
REGION_CASES = {
    "lifting_colour_region": dict(cloud="dense", n=60_000, seed=41, transform=2, qp=34, chroma=-1, subnode=1, search_range=50000,
                                  region=REGION),
    "lifting_refl_region": dict(cloud="lidar", n=50_000, seed=42, transform=2, qp=28, chroma=0, subnode=1, search_range=2500,
                                region=((0, 0, 0), (120_000, 140_000, 90_000), (5, 0))),
    "pred_dense_ctc_region": dict(transform=1, pred_case="dense_ctc", region=REGION),
    # round 5: RAHT slices with a QP region stay on the device too (gpcc_raht_encode_attr_packed_regions /
    # gpcc_raht_decode_attr_regions: the offsets per point derived from the positions)
    "raht_colour_region": dict(cloud="dense", n=60_000, seed=43, transform=0, qp=34, chroma=-1, subnode=1, search_range=50000,
                               region=REGION),
    "raht_refl_sub0_region": dict(cloud="lidar", n=50_000, seed=44, transform=0, qp=28, chroma=0, subnode=0, search_range=2500,
                                  region=((0, 0, 0), (120_000, 140_000, 90_000), (5, 0))),
}


@needs3
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(REGION_CASES))
def test_operator_with_a_qp_region_stays_on_the_device(name):
    case = dict(REGION_CASES[name], lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=True)
    md5, ln, rec = unmodified(case)
    # (the region changes the payload: the case is not vacuous)
    md5_plain, _, _ = unmodified({k: v for k, v in case.items() if k != "region"})
    assert md5 != md5_plain
    assert got["payload_len"] == ln and got["payload_md5"] == md5, "attribute payload differs from the unmodified build"
    assert got["rec_enc_md5"] == rec and got["rec_dec_md5"] == rec
    assert (got["enc_device"], got["enc_cpu"], got["dec_device"], got["dec_cpu"]) == (1, 0, 1, 0)


@needs3
def test_qp_region_on_a_cpu_box():
    from mpeg_pcc_tmc13_amd import _lib
    if _lib.load().gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    case = dict(REGION_CASES["lifting_colour_region"], n=5000, lib="libtmc3_shim3.so")
    got, err = run_worker(case, strict=False)
    md5, ln, rec = unmodified(case)
    assert (got["payload_md5"], got["payload_len"], got["rec_enc_md5"], got["rec_dec_md5"]) == (md5, ln, rec, rec)

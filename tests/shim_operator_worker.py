"""Runs the reference's whole attribute operator (AttributeEncoder::encode +
AttributeDecoder::decode) out of oracle/_ref/libtmc3_shim.so -- the reference
objects with the two link seams replaced by the shim TUs, i.e. with the MI355X
library inside -- and prints what tests/test_shim_operator.py compares with
the unmodified build: md5 of the payload (the attribute brick of the
bitstream) and of the two reconstructions, and the shims' call counters.

    python tests/shim_operator_worker.py <case json>

A process of its own so that only ONE copy of the reference's symbols is ever
loaded next to the HIP library.  TEST INFRASTRUCTURE."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make_case(case):
    from mpeg_pcc_tmc13_amd import lod_params, raht_params, synth
    if case["cloud"] == "dense":
        xyz, attrs = synth.dense_cloud(case["n"], seed=case["seed"], bits=case.get("bits", 9))
    else:
        xyz, attrs = synth.lidar_cloud(case["n"], seed=case["seed"])
    rp = raht_params(qp=case["qp"], chroma_offset=case["chroma"], subnode=bool(case["subnode"]),
                     haar=bool(case.get("haar", 0)), search_range=case["search_range"])
    lp = lod_params(lifting=case["transform"] == 2) if case["transform"] else lod_params()
    if case.get("scalable"):
        lp.scalable_lifting_enabled_flag = 1
        lp.max_neigh_range_minus1 = case.get("neigh_range", 5)
    return xyz, attrs, rp, lp


def pred_case(case):
    """a case of tests/test_oracle_pred.py (transform == 1): cloud, LoD and tool parameters"""
    from mpeg_pcc_tmc13_amd import pred_params
    import test_oracle_pred as top
    xyz, attrs, lp, qp, bitdepth, thr, po = top.make(case["pred_case"])
    pp = pred_params([attrs.shape[0]], qp=qp, chroma_offset=0, bitdepth=bitdepth, threshold=thr,
                     max_levels=lp.num_detail_levels_minus1 + 1, **po)
    return xyz, attrs, lp, pp, thr, qp


def inter_case(case):
    """a slice with attribute inter prediction (transform 1 / 2, one component): the frame is the cloud
    jittered with a tenth of its points gone"""
    from mpeg_pcc_tmc13_amd import lod_params, synth
    if case["cloud"] == "dense":
        xyz, attrs = synth.dense_cloud(case["n"], seed=case["seed"], bits=case.get("bits", 9))
    else:
        xyz, attrs = synth.lidar_cloud(case["n"], seed=case["seed"])
    attrs = attrs[:, :1].copy()
    if attrs.max() > 255:
        attrs = attrs >> 8
    rng = np.random.default_rng(case["seed"])
    keep = rng.random(len(xyz)) > 0.1
    xr = np.clip(xyz + rng.integers(-2, 3, size=xyz.shape), 0, None)[keep].astype(np.int32)
    ar = np.clip(attrs + rng.integers(-6, 7, size=attrs.shape), 0, 255)[keep].astype(np.int32)
    lp = lod_params(lifting=case["transform"] == 2, intra_range=0 if case["transform"] == 2 else 64)
    if case["transform"] == 1:
        lp.intra_lod_prediction_skip_layers = 0
    return xyz, attrs, xr, ar, lp


def two_attr_case(case):
    """two attributes of one slice whose parameter sets pass AttributeLods::isReusable although the
    structures they ask for differ: A = colour, B = reflectance"""
    from mpeg_pcc_tmc13_amd import lod_params, synth
    xyz, col = synth.dense_cloud(case["n"], seed=case["seed"], bits=case.get("bits", 9))
    refl = ((col[:, 0].astype(np.int64) * 3 + xyz[:, 2]) % 256).astype(np.int32)
    ta, tb = case["transforms"]
    lps = []
    for t, kw in ((ta, case.get("lod_a", {})), (tb, case.get("lod_b", {}))):
        # (intra_lod_prediction_skip_layers stays "all": a lifting parameter set cannot say anything else,
        # and isReusable compares it)
        lp = lod_params(lifting=t == 2, blend=bool(case.get("blending", 1)))
        for k, v in kw.items():
            setattr(lp, k, v)
        lps.append(lp)
    return xyz, col, refl, lps[0], ta, lps[1], tb


def multi_slice_case(case):
    """slices of different size of one frame, two attributes each (as two_attr_case)"""
    from mpeg_pcc_tmc13_amd import lod_params, synth
    parts = []
    for i, n in enumerate(case["sizes"]):
        xyz, col = synth.dense_cloud(n, seed=case["seed"] + i, bits=case.get("bits", 9))
        parts.append((xyz, col, ((col[:, 0].astype(np.int64) * 3 + xyz[:, 2]) % 256).astype(np.int32)))
    offs = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.int32)
    ta, tb = case["transforms"]
    lps = [lod_params(lifting=t == 2, blend=bool(case.get("blending", 1))) for t in (ta, tb)]
    return (offs, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]), lps[0], ta, lps[1], tb)


def digest(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    import conftest  # noqa: F401  (loads the package under its alias)
    import lod_helpers as lh
    case = json.loads(sys.argv[1])
    import torch  # noqa: F401  (one HIP runtime per process: torch's, see _lib.load)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", case.get("lib", "libtmc3_shim.so")))
    import time
    t0 = time.time()
    if case.get("region"):
        lh.ref_set_qp_region(case["region"], lib=lib)
    extra = {}
    if case.get("multi_slice"):
        offs, xyz, col, refl, lpa, ta, lpb, tb = multi_slice_case(case)
        payload, lens, enc2, dec2, reused = lh.ref_multi_slice_roundtrip(lpa, ta, lpb, tb, case["qp"], offs, xyz, col, refl, lib=lib)
        rec_enc = np.concatenate([enc2[0].reshape(-1), enc2[1]])
        rec_dec = np.concatenate([dec2[0].reshape(-1), dec2[1]])
        extra = {"reused": list(reused), "payload_lens": lens}
    elif case.get("two_attr"):
        xyz, col, refl, lpa, ta, lpb, tb = two_attr_case(case)
        payload, enc2, dec2, reused = lh.ref_two_attr_roundtrip(lpa, ta, lpb, tb, case["qp"], xyz, col, refl, lib=lib)
        rec_enc = np.concatenate([enc2[0].reshape(-1), enc2[1]])
        rec_dec = np.concatenate([dec2[0].reshape(-1), dec2[1]])
        extra = {"reused": list(reused)}
    elif case.get("inter"):
        xyz, attrs, xr, ar, lp = inter_case(case)
        t0 = time.time()
        payload, rec_enc, rec_dec = lh.ref_inter_roundtrip(lp, case["transform"], case["qp"], 8, case.get("direct", 3), xyz, attrs,
                                                           xr, ar, case.get("search_range", 128), 1, threshold=4, lib=lib)
    elif case["transform"] == 1:
        xyz, attrs, lp, pp, thr, qp = pred_case(case)
        t0 = time.time()
        payload, rec_enc, rec_dec, _ = lh.ref_pred_roundtrip(lp, pp, thr, qp, 0, xyz, attrs, lib=lib)
    else:
        xyz, attrs, rp, lp = make_case(case)
        payload, rec_enc, rec_dec = lh.ref_operator_roundtrip(
            lp, case["transform"], rp, case["qp"], case["chroma"], 8 if attrs.shape[1] == 3 else case.get("bitdepth", 8),
            1, xyz, attrs, lib=lib)
    seconds = time.time() - t0
    if case.get("repeat"):  # a second round trip in the same process: contexts and arenas warm
        t0 = time.time()
        lh.ref_operator_roundtrip(lp, case["transform"], rp, case["qp"], case["chroma"], 8, 1, xyz, attrs, lib=lib)
        seconds = time.time() - t0
    raht, lod, enc, dec = ((C.c_longlong * 2)() for _ in range(4))
    if hasattr(lib, "gpcc_shim_raht_counters"):  # (not in the unmodified build, libtmc3_ref.so)
        lib.gpcc_shim_raht_counters(raht)
        lib.gpcc_shim_lod_counters(lod)
    if hasattr(lib, "gpcc_shim_encoder_counters"):  # seam 3 (libtmc3_shim3.so)
        lib.gpcc_shim_encoder_counters(enc)
        lib.gpcc_shim_decoder_counters(dec)
    print(json.dumps({**extra, "payload_md5": hashlib.md5(payload).hexdigest(), "payload_len": len(payload),
                      "rec_enc_md5": digest(rec_enc), "rec_dec_md5": digest(rec_dec), "seconds": round(seconds, 4),
                      "raht_device": raht[0], "raht_cpu": raht[1], "lod_device": lod[0], "lod_cpu": lod[1],
                      "enc_device": enc[0], "enc_cpu": enc[1], "dec_device": dec[0], "dec_cpu": dec[1]}))


if __name__ == "__main__":
    main()

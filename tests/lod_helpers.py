"""ctypes access to the LoD / lifting entry points of the two CPU checkers
(TEST INFRASTRUCTURE)."""
import ctypes as C

import numpy as np

import oracle_loader as ol

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def oracle_lod_generate(xyz, lp, raw=False):
    return _lod_generate(ol.oracle().lib, "oracle_lod_generate", xyz, lp, raw)


def ref_lod_generate(xyz, lp, raw=False):
    """reference AttributeLods::generate (raw: buildPredictorsFast only) ->
    dict(nc [n], ni [n,3], w [n,3] uint64, indexes [n], npl [L])"""
    return _lod_generate(ol.ref().lib, "ref_lod_generate", xyz, lp, raw)


def _lod_generate(lib, fname, xyz, lp, raw):
    fn = getattr(lib, fname)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, i32p, C.c_int32, C.c_int32, i32p, i32p, u64p, i32p, i32p,
                                     C.POINTER(C.c_int32)]
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    n = len(xyz)
    nc = np.zeros(n, np.int32)
    ni = np.zeros((n, 3), np.int32)
    w = np.zeros((n, 3), np.uint64)
    idx = np.zeros(n, np.int32)
    npl = np.zeros(32, np.int32)
    nl = C.c_int32()
    rc = fn(C.addressof(lp), xyz.reshape(-1), n, int(raw), nc, ni.reshape(-1), w.reshape(-1),
            idx, npl, C.byref(nl))
    assert rc == 0, rc
    return dict(nc=nc, ni=ni, w=w, indexes=idx, npl=npl[:nl.value].copy())


def oracle_lod_generate_inter(xyz, xyz_ref, lp, search_range, frame_distance=1, raw=False):
    return _lod_generate_inter(ol.oracle().lib, "oracle_lod_generate_inter", xyz, xyz_ref, lp, search_range,
                               frame_distance, raw)


def ref_lod_generate_inter(xyz, xyz_ref, lp, search_range, frame_distance=1, raw=False):
    """reference AttributeLods::generate with attribute inter prediction: dict as ref_lod_generate
    plus ref [n,3] (PCCNeighborInfo::interFrameRef; ni is then a point index of the reference frame)"""
    return _lod_generate_inter(ol.ref().lib, "ref_lod_generate_inter", xyz, xyz_ref, lp, search_range,
                               frame_distance, raw)


def _lod_generate_inter(lib, fname, xyz, xyz_ref, lp, search_range, frame_distance, raw):
    fn = getattr(lib, fname)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, i32p, C.c_int32, i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, u64p,
                   i32p, i32p, C.POINTER(C.c_int32), i32p]
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    xyz_ref = np.ascontiguousarray(xyz_ref, dtype=np.int32)
    n = len(xyz)
    nc = np.zeros(n, np.int32)
    ni = np.zeros((n, 3), np.int32)
    w = np.zeros((n, 3), np.uint64)
    idx = np.zeros(n, np.int32)
    npl = np.zeros(32, np.int32)
    ref = np.zeros((n, 3), np.int32)
    nl = C.c_int32()
    rc = fn(C.addressof(lp), xyz.reshape(-1), n, xyz_ref.reshape(-1), len(xyz_ref), search_range, frame_distance,
            int(raw), nc, ni.reshape(-1), w.reshape(-1), idx, npl, C.byref(nl), ref.reshape(-1))
    assert rc == 0, rc
    return dict(nc=nc, ni=ni, w=w, indexes=idx, npl=npl[:nl.value].copy(), ref=ref)


def compute_weights(checker, nc, dist2):
    f = checker.fn("compute_weights", None, [C.c_int32, i32p, u64p])
    nc2 = np.ascontiguousarray(nc, dtype=np.int32).copy()
    w = np.ascontiguousarray(dist2, dtype=np.uint64).copy()
    f(len(nc2), nc2, w.reshape(-1))
    return nc2, w


def lift(checker, forward, lf, lod, attrs, coeffs=None, lcp=None, qp_off=None):
    """-> (coeffs [n,c] coding order, recon [n,c] point order, lcp int8[32])"""
    name = "lift_forward" if forward else "lift_inverse"
    f = checker.fn(name, C.c_int, [C.c_void_p, C.c_int32, C.c_int32, i32p, i32p, i32p, i32p, C.c_void_p, i32p,
                                   i32p, C.c_void_p])
    n = len(lod["nc"])
    a = np.ascontiguousarray(attrs, dtype=np.int32).copy() if forward else np.zeros_like(np.asarray(attrs, np.int32))
    c = a.shape[1]
    co = np.zeros((n, c), np.int32) if forward else np.ascontiguousarray(coeffs, dtype=np.int32).copy()
    l = (C.c_int8 * 32)()
    if lcp is not None:
        for i in range(32):
            l[i] = int(lcp[i])
    q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
    rc = f(C.addressof(lf), n, c, lod["nc"], np.ascontiguousarray(lod["ni"]).reshape(-1),
           np.ascontiguousarray(lod["w"].astype(np.int32)).reshape(-1), lod["indexes"],
           q.ctypes.data_as(C.c_void_p) if q is not None else None, a.reshape(-1), co.reshape(-1), l)
    assert rc == 0
    return co, a, np.array(list(l), dtype=np.int8)


def lift_inter(checker, forward, lf, lod, attrs, attrs_ref, coeffs=None):
    """reflectance lifting with neighbours in a reference frame (lod["ref"], lod["ni"] as
    *_lod_generate_inter give them) -> (coeffs [n,1] coding order, recon [n,1] point order)"""
    f = checker.fn("lift_forward_inter" if forward else "lift_inverse_inter", C.c_int,
                   [C.c_void_p, C.c_int32, i32p, i32p, i32p, i32p, i32p, i32p, i32p, C.c_int32, i32p])
    n = len(lod["nc"])
    a = np.ascontiguousarray(attrs, dtype=np.int32).copy() if forward else np.zeros((n, 1), np.int32)
    co = np.zeros((n, 1), np.int32) if forward else np.ascontiguousarray(coeffs, dtype=np.int32).copy()
    ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
    rc = f(C.addressof(lf), n, lod["nc"], np.ascontiguousarray(lod["ni"]).reshape(-1),
           np.ascontiguousarray(lod["w"].astype(np.int32)).reshape(-1),
           np.ascontiguousarray(lod["ref"], dtype=np.int32).reshape(-1), lod["indexes"], a.reshape(-1), ar, len(ar),
           co.reshape(-1))
    assert rc == 0, rc
    return co, a


def pred_inter(forward, pp, lod, attrs_ref, attrs=None, values=None):
    """the reflectance predicting transform with neighbours in a reference frame
    -> (values [n,1] coding order, recon [n,1] point order, modes [n])"""
    o = ol.oracle()
    f = o.fn("pred_forward_inter" if forward else "pred_inverse_inter", C.c_int,
             [C.c_void_p, C.c_int32, i32p, i32p, i32p, i32p, i32p, i32p, i32p, C.c_int32, i32p, i32p])
    n = len(lod["nc"])
    a = np.ascontiguousarray(attrs, dtype=np.int32).copy() if forward else np.zeros((n, 1), np.int32)
    v = np.zeros((n, 1), np.int32) if forward else np.ascontiguousarray(values, dtype=np.int32).copy()
    ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
    modes = np.zeros(n, np.int32)
    rc = f(C.addressof(pp), n, lod["nc"], np.ascontiguousarray(lod["ni"]).reshape(-1),
           np.ascontiguousarray(lod["w"].astype(np.int32)).reshape(-1),
           np.ascontiguousarray(lod["ref"], dtype=np.int32).reshape(-1), lod["indexes"], a.reshape(-1), ar, len(ar),
           v.reshape(-1), modes)
    assert rc == 0, rc
    return v, a, modes


def ref_inter_roundtrip(lp, transform, qp, bitdepth, direct, xyz, attrs, xyz_ref, attrs_ref, search_range,
                        frame_distance=1, threshold=64, lib=None):
    """the reference operator (encode + decode) with attribute inter prediction, one component
    -> (payload, recon_enc, recon_dec).  `lib`: another build of the same harness (libtmc3_shim.so)."""
    lib = lib or ol.ref().lib
    lib.ref_inter_roundtrip.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, i32p,
                                        C.c_int32, i32p, i32p, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, u8p, C.c_int32]
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    xyz_ref = np.ascontiguousarray(xyz_ref, dtype=np.int32)
    a = np.ascontiguousarray(attrs, dtype=np.int32).reshape(-1)
    ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
    n = len(xyz)
    re = np.zeros(n, np.int32)
    rd = np.zeros(n, np.int32)
    pay = np.zeros(n * 8 + 4096, np.uint8)
    ln = lib.ref_inter_roundtrip(C.addressof(lp), transform, qp, bitdepth, direct, threshold, xyz.reshape(-1), a, n,
                                 xyz_ref.reshape(-1), ar, len(xyz_ref), search_range, frame_distance, re, rd, pay, pay.size)
    return pay[:ln].tobytes(), re.reshape(n, 1), rd.reshape(n, 1)


def ref_operator_roundtrip(lp, transform, rp, qp, chroma, bitdepth, lcp, xyz, attrs, lib=None):
    """reference AttributeEncoder::encode + AttributeDecoder::decode ->
    (payload bytes, recon_enc, recon_dec).  `lib`: another build of the same
    harness (oracle/_ref/libtmc3_shim.so: the operator with the device inside)."""
    lib = lib or ol.ref().lib
    lib.ref_operator_roundtrip.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, i32p, i32p, C.c_int32, C.c_int32, i32p, i32p, u8p, C.c_int32]
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    attrs = np.ascontiguousarray(attrs, dtype=np.int32)
    n, c = attrs.shape
    re = np.zeros(n * c, np.int32)
    rd = np.zeros(n * c, np.int32)
    pay = np.zeros(n * c * 8 + 4096, np.uint8)
    ln = lib.ref_operator_roundtrip(C.addressof(lp), transform, C.addressof(rp), qp, chroma, bitdepth, int(lcp),
                                    xyz.reshape(-1), attrs.reshape(-1), n, c, re, rd, pay, pay.size)
    return pay[:ln].tobytes(), re.reshape(n, c), rd.reshape(n, c)


def ref_set_qp_region(region, lib=None):
    """the QP region of the attribute brick headers the harness builds from here on: None, or
    (origin xyz, size xyz, (qp offset luma, chroma)) -- regionOrigin / regionSize / attr_region_qp_offset"""
    lib = lib or ol.ref().lib
    lib.ref_set_qp_region.argtypes = [C.c_int32, i32p]
    lib.ref_set_qp_region.restype = None
    flat = np.zeros(8, np.int32)
    if region is not None:
        flat[:] = list(region[0]) + list(region[1]) + list(region[2])
    lib.ref_set_qp_region(int(region is not None), flat)


def ref_two_attr_roundtrip(lp_a, transform_a, lp_b, transform_b, qp, xyz, colours, refl, lib=None):
    """colour (parameter set A) then reflectance (B) of one slice through the operator the way the
    reference's encoder / decoder drive it: the same coder object serves B when isReusable(B) says so.
    -> (both payloads back to back, (rec colour, rec reflectance) of the encoder, of the decoder,
    (encoder object kept, decoder object kept))"""
    lib = lib or ol.ref().lib
    lib.ref_two_attr_roundtrip.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, i32p, i32p, i32p,
                                           C.c_int32, i32p, i32p, i32p, i32p, u8p, C.c_int32, i32p]
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    colours = np.ascontiguousarray(colours, dtype=np.int32)
    refl = np.ascontiguousarray(refl, dtype=np.int32).reshape(-1)
    n = len(xyz)
    out = [np.zeros(3 * n, np.int32), np.zeros(n, np.int32), np.zeros(3 * n, np.int32), np.zeros(n, np.int32)]
    pay = np.zeros(n * 4 * 8 + 8192, np.uint8)
    reused = np.zeros(2, np.int32)
    ln = lib.ref_two_attr_roundtrip(C.addressof(lp_a), transform_a, C.addressof(lp_b), transform_b, qp, xyz.reshape(-1),
                                    colours.reshape(-1), refl, n, out[0], out[1], out[2], out[3], pay, pay.size, reused)
    assert 0 < ln <= pay.size
    return pay[:ln].tobytes(), (out[0].reshape(n, 3), out[1]), (out[2].reshape(n, 3), out[3]), tuple(int(v) for v in reused)


def ref_multi_slice_roundtrip(lp_a, transform_a, lp_b, transform_b, qp, offsets, xyz, colours, refl, lib=None):
    """several slices of a frame through the operator as the reference's compressPartition drives it: new coder
    objects per slice, colour then reflectance inside a slice, the entropy context memory of each attribute carried
    from slice to slice.  -> (payloads back to back, lengths [2 per slice], (rec colour, rec reflectance) of the
    encoder, of the decoder, objects kept for B per slice)"""
    lib = lib or ol.ref().lib
    lib.ref_multi_slice_roundtrip.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, i32p, i32p,
                                              i32p, i32p, i32p, i32p, i32p, i32p, u8p, C.c_int32, i32p, i32p]
    offs = np.ascontiguousarray(offsets, dtype=np.int32)
    k = len(offs) - 1
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    colours = np.ascontiguousarray(colours, dtype=np.int32)
    refl = np.ascontiguousarray(refl, dtype=np.int32).reshape(-1)
    n = len(xyz)
    out = [np.zeros(3 * n, np.int32), np.zeros(n, np.int32), np.zeros(3 * n, np.int32), np.zeros(n, np.int32)]
    pay = np.zeros(n * 4 * 8 + 8192 * k, np.uint8)
    lens = np.zeros(2 * k, np.int32)
    reused = np.zeros(k, np.int32)
    ln = lib.ref_multi_slice_roundtrip(C.addressof(lp_a), transform_a, C.addressof(lp_b), transform_b, qp, k, offs,
                                       xyz.reshape(-1), colours.reshape(-1), refl, out[0], out[1], out[2], out[3], pay,
                                       pay.size, lens, reused)
    assert 0 < ln <= pay.size
    return (pay[:ln].tobytes(), [int(v) for v in lens], (out[0].reshape(n, 3), out[1]), (out[2].reshape(n, 3), out[3]),
            [int(v) for v in reused])


def oracle_estimate_dist2(xyz, period=100, search_range=128, percentile=0.85):
    lib = ol.oracle().lib
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    lib.oracle_estimate_dist2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    return lib.oracle_estimate_dist2(xyz.ctypes.data, xyz.shape[0], period, search_range, C.c_float(percentile), None)


def ref_estimate_dist2(xyz, period=100, search_range=128, percentile=0.85):
    lib = ol.ref().lib
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    lib.ref_estimate_dist2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float]
    return lib.ref_estimate_dist2(xyz.ctypes.data, xyz.shape[0], period, search_range, C.c_float(percentile))


def ref_last_abh_size():
    """size of the brick header at the start of the last ref_operator_roundtrip payload"""
    return ol.ref().lib.ref_last_abh_size()


def oracle_zero_run_pack(coeffs, n, c, planar):
    """-> (runs [m], values [m, c], trailing_run)"""
    lib = ol.oracle().lib
    co = np.ascontiguousarray(coeffs, dtype=np.int32).reshape(-1)
    runs = np.zeros(n, np.int32)
    vals = np.zeros(n * c, np.int32)
    tr = C.c_int32()
    lib.oracle_zero_run_pack.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, C.POINTER(C.c_int32)]
    m = lib.oracle_zero_run_pack(co, n, c, int(planar), runs, vals, C.byref(tr))
    return runs[:m].copy(), vals[:m * c].reshape(m, c).copy(), tr.value


_entropy = {}


def entropy_available():
    import os
    return os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "libtmc3_entropy.so"))


def ref_entropy_encode_symbols(c, num_points, runs, values, trailing):
    """the reference's own PCCResidualsEncoder on a symbol stream -> arithmetic-coded bytes"""
    import os
    if "lib" not in _entropy:
        _entropy["lib"] = C.CDLL(os.path.join(ol.ORACLE_DIR, "_ref", "libtmc3_entropy.so"))
    lib = _entropy["lib"]
    runs = np.ascontiguousarray(runs, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.int32).reshape(-1)
    out = np.zeros(num_points * 3 * 2 + 2048, np.uint8)
    lib.ref_entropy_encode_symbols.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_int32, C.c_int32, u8p, C.c_int32]
    ln = lib.ref_entropy_encode_symbols(c, num_points, runs if len(runs) else np.zeros(1, np.int32),
                                        values if len(values) else np.zeros(1, np.int32), len(runs), trailing,
                                        out, out.size)
    assert ln >= 0
    return out[:ln].tobytes()


def oracle_binarise_symbols(runs, values, trailing, c):
    """oracle/symbols_oracle.c: the decisions of PCCResidualsEncoder for a symbol stream -> uint8 [(ctx << 1) | bin]"""
    lib = ol.oracle().lib
    runs = np.ascontiguousarray(runs, dtype=np.int32)
    vals = np.ascontiguousarray(values, dtype=np.int32).reshape(-1)
    lib.oracle_binarise_symbols.restype = C.c_int64
    lib.oracle_binarise_symbols.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
    n = lib.oracle_binarise_symbols(runs.ctypes.data, vals.ctypes.data, len(runs), int(trailing), c, None, 0)
    out = np.zeros(max(n, 1), np.uint8)
    lib.oracle_binarise_symbols(runs.ctypes.data, vals.ctypes.data, len(runs), int(trailing), c, out.ctypes.data, n)
    return out[:n]


def ref_entropy_encode_bins(bins, num_points):
    """the reference's arithmetic coder + context models on a decision stream -> arithmetic-coded bytes"""
    import os
    if "lib" not in _entropy:
        _entropy["lib"] = C.CDLL(os.path.join(ol.ORACLE_DIR, "_ref", "libtmc3_entropy.so"))
    lib = _entropy["lib"]
    bins = np.ascontiguousarray(bins, dtype=np.uint8)
    out = np.zeros(num_points * 3 * 2 + 2048, np.uint8)
    lib.ref_entropy_encode_bins.argtypes = [C.c_void_p, C.c_int64, C.c_int32, u8p, C.c_int32]
    ln = lib.ref_entropy_encode_bins(bins.ctypes.data, len(bins), num_points, out, out.size)
    assert ln >= 0
    return out[:ln].tobytes()


# ---- predicting transform ---------------------------------------------------
def oracle_pred(forward, pp, lod, attrs=None, values=None, icp=None, qp_off=None):
    """oracle/pred_oracle.c -> (values [n,c] coding order, recon [n,c] point order,
    icp int8 [32,3], modes [n] (-1 = not eligible))"""
    lib = ol.oracle().lib
    f = lib.oracle_pred_forward if forward else lib.oracle_pred_inverse
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, i32p, i32p, i32p, i32p, C.c_void_p, i32p, i32p, C.c_void_p, i32p]
    n = len(lod["nc"])
    if forward:
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        c = a.shape[1]
        v = np.zeros((n, c), np.int32)
        l = np.zeros((32, 3), np.int8)
    else:
        v = np.ascontiguousarray(values, dtype=np.int32).copy()
        c = v.shape[1]
        a = np.zeros((n, c), np.int32)
        l = np.zeros((32, 3), np.int8) if icp is None else np.ascontiguousarray(icp, dtype=np.int8).copy()
    modes = np.zeros(n, np.int32)
    q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
    rc = f(C.addressof(pp), n, c, lod["nc"], np.ascontiguousarray(lod["ni"]).reshape(-1),
           np.ascontiguousarray(lod["w"].astype(np.int32)).reshape(-1), lod["indexes"],
           q.ctypes.data_as(C.c_void_p) if q is not None else None, a.reshape(-1), v.reshape(-1),
           l.ctypes.data_as(C.c_void_p), modes)
    assert rc == 0
    return v, a, l, modes


def ref_pred_roundtrip(lp, pp, aps_threshold, qp, chroma, xyz, attrs, lib=None):
    """reference AttributeEncoder::encode + AttributeDecoder::decode for the
    predicting transform -> (payload, recon_enc, recon_dec, icp int8 [32,3]).
    `lib`: another build of the same harness (libtmc3_shim3.so)."""
    lib = lib or ol.ref().lib
    lib.ref_pred_roundtrip.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, C.c_int32,
                                       C.c_int32, i32p, i32p, u8p, C.c_int32, C.c_void_p]
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    attrs = np.ascontiguousarray(attrs, dtype=np.int32)
    n, c = attrs.shape
    re = np.zeros(n * c, np.int32)
    rd = np.zeros(n * c, np.int32)
    pay = np.zeros(n * c * 8 + 4096, np.uint8)
    icp = np.zeros((32, 3), np.int8)
    ln = lib.ref_pred_roundtrip(C.addressof(lp), C.addressof(pp), aps_threshold, qp, chroma, xyz.reshape(-1),
                                attrs.reshape(-1), n, c, re, rd, pay, pay.size, icp.ctypes.data_as(C.c_void_p))
    return pay[:ln].tobytes(), re.reshape(n, c), rd.reshape(n, c), icp


_entropy_dec = None


def entropy_dec_available():
    import os
    return os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "libtmc3_entropy_dec.so"))


def ref_entropy_decode_symbols(payload, n, c):
    """the reference's PCCResidualsDecoder over the arithmetic-coded part of a
    payload -> values [n,c] in coding order (zero runs expanded)"""
    global _entropy_dec
    import os
    if _entropy_dec is None:
        _entropy_dec = C.CDLL(os.path.join(ol.ORACLE_DIR, "_ref", "libtmc3_entropy_dec.so"))
        _entropy_dec.ref_entropy_decode_symbols.argtypes = [C.c_int32, C.c_int32, u8p, C.c_int32, i32p]
    buf = np.frombuffer(payload, dtype=np.uint8).copy()
    out = np.zeros(n * c, np.int32)
    rc = _entropy_dec.ref_entropy_decode_symbols(c, n, buf, len(buf), out)
    assert rc == 0
    return out.reshape(n, c)

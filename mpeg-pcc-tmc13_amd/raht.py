"""Host-side mirror of the reference's RAHT interface over the C ABI.

``raht_forward`` / ``raht_inverse`` have the argument meaning of
``pcc::regionAdaptiveHierarchicalTransform`` / ``...InverseTransform``
(reference tmc3/RAHT.h:47-69): ascending Morton codes, row-major attributes
(overwritten with the reconstruction), planar coefficients.  Errors raise
(the reference asserts); GPCC_ERR_UNSUPPORTED tells the caller to keep the
slice on the reference CPU path."""
import ctypes as C

import numpy as np

from . import _lib
from .params import RahtParams


class Context:
    """gpcc_ctx: one HIP device + stream + workspace."""

    def __init__(self, device=0, stream=None):
        """stream: None = the library creates its own stream; a hipStream_t
        handle (e.g. torch.cuda.Stream().cuda_stream) = run there; 0 = the
        legacy default stream (GPCC_STREAM_LEGACY), which is what a torch
        default stream's handle means."""
        self._lib = _lib.load()
        h = C.c_void_p()
        handle = 0 if stream is None else (1 if stream == 0 else stream)
        _lib.check(self._lib.gpcc_ctx_create(device, C.c_void_p(handle), C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpcc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(self._lib.gpcc_ctx_synchronize(self._h))

    def workspace_bytes(self):
        return int(self._lib.gpcc_ctx_workspace_bytes(self._h))

    def set_morton_bits(self, bits):
        _lib.check(self._lib.gpcc_ctx_set_morton_bits(self._h, int(bits)))

    def reserve(self, max_points, max_slices=1, max_c=3):
        """gpcc_ctx_reserve: everything the RAHT entries allocate on demand, ahead of the first call"""
        _lib.check(self._lib.gpcc_ctx_reserve(self._h, int(max_points), int(max_slices), int(max_c)))

    def set_fast_arith(self, on):
        """doubles where they are exact (default) / int64 always in the sub-node kernels"""
        _lib.check(self._lib.gpcc_ctx_set_fast_arith(self._h, int(bool(on))))

    def pred_pass_stats(self):
        """{slices, passes, most_passes, declined_at_the_limit} of the predicting encoder's fixed-point
        iteration (direct predictors) since the context was created"""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib.gpcc_ctx_pred_pass_stats(self._h, out))
        return dict(zip(("slices", "passes", "most_passes", "declined_at_the_limit"), (int(v) for v in out)))

    def set_profiling(self, on):
        _lib.check(self._lib.gpcc_ctx_set_profiling(self._h, int(bool(on))))

    def stats(self):
        """gpcc_ctx_stats: {calls_ok, calls_unsupported, calls_failed, points_ok}."""
        st = _lib.CtxStats()
        _lib.check(self._lib.gpcc_ctx_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in st._fields_}

    def kernel_times(self):
        """{kernel name: (total ms, launches)} since the last call."""
        buf = (_lib.KernelTime * 256)()
        n = self._lib.gpcc_ctx_kernel_times(self._h, buf, 256)
        if n < 0:
            _lib.check(n)
        return {buf[i].name.decode(): (buf[i].total_ms, buf[i].launches) for i in range(n)}

    # ---- host tier ------------------------------------------------------
    def raht_forward(self, params: RahtParams, morton, attrs, qp_off=None):
        """-> (coeffs int32 [c*n] planar, recon int32 [n, c])"""
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        rec = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = rec.shape
        coeffs = np.zeros(c * n, dtype=np.int32)
        q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_forward(
            self._h, C.byref(params), morton.ctypes.data, q.ctypes.data if q is not None else None,
            rec.ctypes.data, coeffs.ctypes.data, n, c))
        return coeffs, rec

    def raht_inverse(self, params: RahtParams, morton, coeffs, c, qp_off=None):
        """-> recon int32 [n, c] (unclipped)"""
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        n = morton.shape[0]
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        rec = np.zeros((n, c), dtype=np.int32)
        q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_inverse(
            self._h, C.byref(params), morton.ctypes.data, q.ctypes.data if q is not None else None,
            rec.ctypes.data, co.ctypes.data, n, c))
        return rec

    def raht_forward_inter(self, params: RahtParams, inter, morton, attrs, morton_ref, attrs_ref, qp_off=None):
        """RAHT with attribute inter prediction (gpcc_raht_forward_inter)
        -> (coeffs int32 [c*n] planar, recon int32 [n, c], layer modes, filter taps)"""
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        rec = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = rec.shape
        mref = np.ascontiguousarray(morton_ref, dtype=np.int64)
        aref = np.ascontiguousarray(attrs_ref, dtype=np.int32)
        coeffs = np.zeros(c * n, dtype=np.int32)
        modes, taps = np.zeros(32, np.int32), np.zeros(32, np.int32)
        nm, nt = C.c_int32(0), C.c_int32(0)
        q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_forward_inter(
            self._h, C.byref(params), C.addressof(inter), morton.ctypes.data, q.ctypes.data if q is not None else None,
            rec.ctypes.data, coeffs.ctypes.data, n, c,
            mref.ctypes.data, aref.ctypes.data, len(mref), modes.ctypes.data, C.byref(nm), taps.ctypes.data, C.byref(nt)))
        return coeffs, rec, modes[:nm.value].copy(), taps[:nt.value].copy()

    def raht_inverse_inter(self, params: RahtParams, inter, morton, coeffs, c, morton_ref, attrs_ref, modes, taps, qp_off=None):
        """-> recon int32 [n, c] (gpcc_raht_inverse_inter)"""
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        n = morton.shape[0]
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        mref = np.ascontiguousarray(morton_ref, dtype=np.int64)
        aref = np.ascontiguousarray(attrs_ref, dtype=np.int32)
        m = np.zeros(32, np.int32)
        t = np.zeros(32, np.int32)
        m[:len(modes)] = modes
        t[:len(taps)] = taps
        rec = np.zeros((n, c), dtype=np.int32)
        q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_inverse_inter(
            self._h, C.byref(params), C.addressof(inter), morton.ctypes.data, q.ctypes.data if q is not None else None,
            rec.ctypes.data, co.ctypes.data, n, c,
            mref.ctypes.data, aref.ctypes.data, len(mref), m.ctypes.data, len(modes), t.ctypes.data, len(taps)))
        return rec

    def morton_sort(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        n = xyz.shape[0]
        morton = np.zeros(n, dtype=np.int64)
        order = np.zeros(n, dtype=np.int32)
        _lib.check(self._lib.gpcc_attr_morton_sort(
            self._h, xyz.ctypes.data, n, morton.ctypes.data, order.ctypes.data))
        return morton, order

    # ---- device tier (raw device addresses, e.g. torch tensor.data_ptr()) -
    def dev_raht_forward(self, params, offsets, d_morton, d_attrs, d_coeffs, c, d_qp_off=None):
        off = (C.c_int64 * len(offsets))(*[int(o) for o in offsets])
        _lib.check(self._lib.gpcc_dev_raht_forward(
            self._h, C.byref(params), len(offsets) - 1, off, C.c_void_p(d_morton),
            C.c_void_p(d_qp_off or 0), C.c_void_p(d_attrs), C.c_void_p(d_coeffs), c))

    def dev_raht_inverse(self, params, offsets, d_morton, d_attrs, d_coeffs, c, d_qp_off=None):
        off = (C.c_int64 * len(offsets))(*[int(o) for o in offsets])
        _lib.check(self._lib.gpcc_dev_raht_inverse(
            self._h, C.byref(params), len(offsets) - 1, off, C.c_void_p(d_morton),
            C.c_void_p(d_qp_off or 0), C.c_void_p(d_attrs), C.c_void_p(d_coeffs), c))

    # ---- lifting transform (predictors given, host tier) -------------------
    def _lift(self, forward, params, nc, ni, nw, indexes, attrs, coeffs, lcp, qp_off):
        nc = np.ascontiguousarray(nc, dtype=np.int32)
        ni = np.ascontiguousarray(ni, dtype=np.int32)
        nw = np.ascontiguousarray(nw, dtype=np.int32)
        ix = np.ascontiguousarray(indexes, dtype=np.int32)
        n = nc.shape[0]
        q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
        l = np.zeros(32, dtype=np.int8) if lcp is None else np.ascontiguousarray(lcp, dtype=np.int8).copy()
        if forward:
            a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
            c = a.shape[1]
            co = np.zeros((n, c), dtype=np.int32)
            fn = self._lib.gpcc_lift_forward
        else:
            co = np.ascontiguousarray(coeffs, dtype=np.int32)
            c = co.shape[1]
            a = np.zeros((n, c), dtype=np.int32)
            fn = self._lib.gpcc_lift_inverse
        _lib.check(fn(self._h, C.byref(params), n, c, nc.ctypes.data, ni.ctypes.data, nw.ctypes.data,
                      ix.ctypes.data, q.ctypes.data if q is not None else None, a.ctypes.data,
                      co.ctypes.data, l.ctypes.data))
        return co, a, l

    def lift_forward(self, params, nc, ni, nw, indexes, attrs, qp_off=None):
        """encodeColorsLift / encodeReflectancesLift minus the entropy calls ->
        (coeffs [n,c] coding order, recon [n,c] point order (clipped), lcp int8[32])"""
        return self._lift(True, params, nc, ni, nw, indexes, attrs, None, None, qp_off)

    def lift_inverse(self, params, nc, ni, nw, indexes, coeffs, lcp=None, qp_off=None):
        """decodeColorsLift / decodeReflectancesLift after the entropy decode -> recon [n,c]"""
        return self._lift(False, params, nc, ni, nw, indexes, None, coeffs, lcp, qp_off)[1]

    def lod_compute_weights(self, nc, dist2):
        """PCCPredictor::computeWeights -> (neighbour counts, 8-bit weights [n,3])"""
        nc = np.ascontiguousarray(nc, dtype=np.int32).copy()
        d = np.ascontiguousarray(dist2, dtype=np.uint64)
        w = np.zeros(d.shape, dtype=np.int32)
        _lib.check(self._lib.gpcc_lod_compute_weights(self._h, nc.shape[0], nc.ctypes.data, d.ctypes.data,
                                                      w.ctypes.data))
        return nc, w

    def lod_build(self, params, xyz):
        """AttributeLods::generate -> dict(nc, ni, w, indexes, npl) in coding order"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        n = xyz.shape[0]
        nc = np.zeros(n, np.int32)
        ni = np.zeros((n, 3), np.int32)
        w = np.zeros((n, 3), np.int32)
        idx = np.zeros(n, np.int32)
        npl = np.zeros(32, np.int32)
        nl = C.c_int32()
        _lib.check(self._lib.gpcc_lod_build(self._h, C.byref(params), xyz.ctypes.data, n, nc.ctypes.data,
                                            ni.ctypes.data, w.ctypes.data, idx.ctypes.data, npl.ctypes.data,
                                            C.byref(nl)))
        return dict(nc=nc, ni=ni, w=w, indexes=idx, npl=npl[:nl.value].copy())

    def lod_build_inter(self, params, xyz, xyz_ref, search_range, frame_distance=1):
        """AttributeLods::generate with attribute inter prediction -> dict as lod_build plus
        ref [n,3] (neighbour lives in the reference frame; ni is then a point index there)"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        xyz_ref = np.ascontiguousarray(xyz_ref, dtype=np.int32)
        n = xyz.shape[0]
        nc = np.zeros(n, np.int32)
        ni = np.zeros((n, 3), np.int32)
        w = np.zeros((n, 3), np.int32)
        idx = np.zeros(n, np.int32)
        npl = np.zeros(32, np.int32)
        ref = np.zeros((n, 3), np.int32)
        nl = C.c_int32()
        _lib.check(self._lib.gpcc_lod_build_inter(self._h, C.byref(params), xyz.ctypes.data, n, xyz_ref.ctypes.data,
                                                  xyz_ref.shape[0], search_range, frame_distance, nc.ctypes.data,
                                                  ni.ctypes.data, w.ctypes.data, idx.ctypes.data, npl.ctypes.data,
                                                  C.byref(nl), ref.ctypes.data))
        return dict(nc=nc, ni=ni, w=w, indexes=idx, npl=npl[:nl.value].copy(), ref=ref)

    def lift_inter(self, forward, params, lod, attrs_ref, attrs=None, coeffs=None):
        """reflectance lifting with neighbours in a reference frame (lod as lod_build_inter returns it)
        -> (coeffs [n,1] coding order, recon [n,1] point order)"""
        n = len(lod["nc"])
        nc = np.ascontiguousarray(lod["nc"], dtype=np.int32)
        ni = np.ascontiguousarray(lod["ni"], dtype=np.int32)
        nw = np.ascontiguousarray(lod["w"], dtype=np.int32)
        xr = np.ascontiguousarray(lod["ref"], dtype=np.int32)
        ix = np.ascontiguousarray(lod["indexes"], dtype=np.int32)
        ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy() if forward else np.zeros((n, 1), np.int32)
        co = np.zeros((n, 1), np.int32) if forward else np.ascontiguousarray(coeffs, dtype=np.int32).copy()
        fn = self._lib.gpcc_lift_forward_inter if forward else self._lib.gpcc_lift_inverse_inter
        _lib.check(fn(self._h, C.byref(params), n, nc.ctypes.data, ni.ctypes.data, nw.ctypes.data, xr.ctypes.data,
                      ix.ctypes.data, a.ctypes.data, ar.ctypes.data, len(ar), co.ctypes.data))
        return co, a

    def pred_inter(self, forward, params, lod, attrs_ref, attrs=None, values=None):
        """the reflectance predicting transform with neighbours in a reference frame (lod as
        lod_build_inter returns it) -> (values [n,1] coding order, recon [n,1] point order)"""
        n = len(lod["nc"])
        nc = np.ascontiguousarray(lod["nc"], dtype=np.int32)
        ni = np.ascontiguousarray(lod["ni"], dtype=np.int32)
        nw = np.ascontiguousarray(lod["w"], dtype=np.int32)
        xr = np.ascontiguousarray(lod["ref"], dtype=np.int32)
        ix = np.ascontiguousarray(lod["indexes"], dtype=np.int32)
        ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy() if forward else np.zeros((n, 1), np.int32)
        v = np.zeros((n, 1), np.int32) if forward else np.ascontiguousarray(values, dtype=np.int32).copy()
        fn = self._lib.gpcc_pred_forward_inter if forward else self._lib.gpcc_pred_inverse_inter
        _lib.check(fn(self._h, C.byref(params), n, nc.ctypes.data, ni.ctypes.data, nw.ctypes.data, xr.ctypes.data,
                      ix.ctypes.data, a.ctypes.data, ar.ctypes.data, len(ar), v.ctypes.data))
        return v, a

    def estimate_dist2(self, xyz, sampling_period=100, search_range=128, percentile=0.85):
        """pcc::estimateDist2 (encoder.cpp:1203 uses period 100, range 128) -> shift bits"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        out = C.c_int32()
        _lib.check(self._lib.gpcc_estimate_dist2(self._h, xyz.ctypes.data, xyz.shape[0], sampling_period,
                                                 search_range, C.c_float(percentile), C.byref(out)))
        return out.value

    def recolour(self, params, src_xyz, src_attrs, tgt_xyz, scale=1.0, offset=(0, 0, 0)):
        """pcc::recolour (pointset_processing.cpp:926): the attributes of the source cloud
        transferred to the target positions -> int32 [nt, c]"""
        sx = np.ascontiguousarray(src_xyz, dtype=np.int32)
        sa = np.ascontiguousarray(src_attrs, dtype=np.int32)
        tx = np.ascontiguousarray(tgt_xyz, dtype=np.int32)
        ns, c = sa.shape
        nt = tx.shape[0]
        out = np.zeros((nt, c), dtype=np.int32)
        off = (C.c_int32 * 3)(*[int(v) for v in offset])
        _lib.check(self._lib.gpcc_recolour(
            self._h, C.byref(params), sx.ctypes.data, sa.ctypes.data, ns, tx.ctypes.data, nt, c,
            C.c_float(scale), off, out.ctypes.data))
        return out

    # ---- whole slice driver (sort + marshal + transform + clip + scatter) ----
    def raht_encode_attr(self, params, xyz, attrs, bitdepth=8):
        """encode{Colors,Reflectances}TransformRaht minus the entropy loop ->
        (coeffs planar [c*n] Morton order, clipped recon [n,c] POINT order)"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = a.shape
        co = np.zeros(c * n, dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_encode_attr(self._h, C.byref(params), xyz.ctypes.data, a.ctypes.data,
                                                   co.ctypes.data, n, c, bitdepth))
        return co, a

    def raht_decode_attr(self, params, xyz, coeffs, c, bitdepth=8):
        """decode{Colors,Reflectances}Raht after the entropy decode -> clipped recon [n,c] POINT order"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        n = xyz.shape[0]
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        a = np.zeros((n, c), dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_decode_attr(self._h, C.byref(params), xyz.ctypes.data, a.ctypes.data,
                                                   co.ctypes.data, n, c, bitdepth))
        return a

    # ---- lifting coder of one slice: LoD build + transform, predictors stay on the device ----
    def lift_encode_attr(self, lod_params, lift_params, xyz, attrs):
        """AttributeLods::generate + encode{Colors,Reflectances}Lift minus the entropy loop ->
        (coeffs [n,c] coding order, clipped recon [n,c] point order, lcp int8[32], indexes [n]);
        lift_params.num_lods / num_points_in_lod are filled in."""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = a.shape
        co = np.zeros((n, c), dtype=np.int32)
        lcp = np.zeros(32, dtype=np.int8)
        idx = np.zeros(n, dtype=np.int32)
        _lib.check(self._lib.gpcc_lift_encode_attr(self._h, C.byref(lod_params), C.byref(lift_params),
                                                   xyz.ctypes.data, a.ctypes.data, co.ctypes.data,
                                                   lcp.ctypes.data, idx.ctypes.data, n, c))
        return co, a, lcp, idx

    def lift_decode_attr(self, lod_params, lift_params, xyz, coeffs, lcp=None):
        """AttributeLods::generate + decode{Colors,Reflectances}Lift after the entropy decode -> recon [n,c]"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        n, c = co.shape
        a = np.zeros((n, c), dtype=np.int32)
        l = np.zeros(32, dtype=np.int8) if lcp is None else np.ascontiguousarray(lcp, dtype=np.int8).copy()
        _lib.check(self._lib.gpcc_lift_decode_attr(self._h, C.byref(lod_params), C.byref(lift_params),
                                                   xyz.ctypes.data, a.ctypes.data, co.ctypes.data,
                                                   l.ctypes.data, None, n, c))
        return a

    # ---- predicting transform ----
    def _pred(self, forward, params, nc, ni, nw, indexes, attrs, values, icp, qp_off):
        nc = np.ascontiguousarray(nc, dtype=np.int32)
        ni = np.ascontiguousarray(ni, dtype=np.int32)
        nw = np.ascontiguousarray(nw, dtype=np.int32)
        ix = np.ascontiguousarray(indexes, dtype=np.int32)
        n = nc.shape[0]
        if forward:
            a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
            c = a.shape[1]
            v = np.zeros((n, c), dtype=np.int32)
            l = np.zeros((32, 3), dtype=np.int8)
        else:
            v = np.ascontiguousarray(values, dtype=np.int32)
            c = v.shape[1]
            a = np.zeros((n, c), dtype=np.int32)
            l = np.zeros((32, 3), dtype=np.int8) if icp is None else np.ascontiguousarray(icp, dtype=np.int8).copy()
        q = None if qp_off is None else np.ascontiguousarray(qp_off, dtype=np.int32)
        f = self._lib.gpcc_pred_forward if forward else self._lib.gpcc_pred_inverse
        _lib.check(f(self._h, C.byref(params), n, c, nc.ctypes.data, ni.ctypes.data, nw.ctypes.data, ix.ctypes.data,
                     q.ctypes.data if q is not None else None, a.ctypes.data, v.ctypes.data, l.ctypes.data))
        return v, a, l

    def pred_forward(self, params, nc, ni, nw, indexes, attrs, qp_off=None):
        """encodeColorsPred / encodeReflectancesPred minus the entropy calls (no direct
        predictors) -> (values [n,c] coding order, recon [n,c] point order, icp int8[32,3])"""
        return self._pred(True, params, nc, ni, nw, indexes, attrs, None, None, qp_off)

    def pred_inverse(self, params, nc, ni, nw, indexes, values, icp=None, qp_off=None):
        """decodeColorsPred / decodeReflectancesPred after the entropy decode -> recon [n,c]"""
        return self._pred(False, params, nc, ni, nw, indexes, None, values, icp, qp_off)[1]

    def pred_encode_attr(self, lod_params, pred_params, xyz, attrs):
        """AttributeLods::generate + encode...Pred minus the entropy loop ->
        (values, recon, icp int8[32,3], indexes); pred_params gets the LoD structure"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = a.shape
        v = np.zeros((n, c), dtype=np.int32)
        l = np.zeros((32, 3), dtype=np.int8)
        idx = np.zeros(n, dtype=np.int32)
        _lib.check(self._lib.gpcc_pred_encode_attr(self._h, C.byref(lod_params), C.byref(pred_params),
                                                   xyz.ctypes.data, a.ctypes.data, v.ctypes.data, l.ctypes.data,
                                                   idx.ctypes.data, n, c))
        return v, a, l, idx

    def pred_decode_attr(self, lod_params, pred_params, xyz, values, icp=None):
        """AttributeLods::generate + decode...Pred after the entropy decode -> recon [n,c]"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.int32)
        n, c = v.shape
        a = np.zeros((n, c), dtype=np.int32)
        l = np.zeros((32, 3), dtype=np.int8) if icp is None else np.ascontiguousarray(icp, dtype=np.int8).copy()
        _lib.check(self._lib.gpcc_pred_decode_attr(self._h, C.byref(lod_params), C.byref(pred_params),
                                                   xyz.ctypes.data, a.ctypes.data, v.ctypes.data, l.ctypes.data,
                                                   None, n, c))
        return a

    # ---- device tier of the LoD build / lifting coder (buffers are raw device pointers) ----
    def dev_lod_build(self, lod_params, offsets, d_xyz, d_count, d_index, d_weight, d_indexes):
        """gpcc_dev_lod_build -> list of cumulative LoD sizes per slice"""
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        s = len(offs) - 1
        npl = np.zeros((s, 32), dtype=np.int32)
        nl = np.zeros(s, dtype=np.int32)
        _lib.check(self._lib.gpcc_dev_lod_build(
            self._h, C.byref(lod_params), s, offs.ctypes.data_as(C.POINTER(C.c_int64)), d_xyz, d_count, d_index,
            d_weight, d_indexes, npl.ctypes.data, nl.ctypes.data))
        return [list(npl[i, :nl[i]]) for i in range(s)]

    def dev_lift_attr(self, encode, lod_params, lift_params_list, offsets, d_xyz, d_attrs, d_coeffs, c, lcp=None,
                      d_indexes=None):
        """gpcc_dev_lift_encode_attr / _decode_attr on device buffers; lift_params_list: one
        LiftParams per slice (filled with the LoD structure); -> lcp int8 [slices, 32]"""
        from .params import LiftParams
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        s = len(offs) - 1
        arr = (LiftParams * s)(*lift_params_list)
        l = np.zeros((s, 32), dtype=np.int8) if lcp is None else np.ascontiguousarray(lcp, dtype=np.int8).copy()
        fn = self._lib.gpcc_dev_lift_encode_attr if encode else self._lib.gpcc_dev_lift_decode_attr
        _lib.check(fn(self._h, C.byref(lod_params), arr, s, offs.ctypes.data_as(C.POINTER(C.c_int64)), d_xyz,
                      d_attrs, d_coeffs, l.ctypes.data, d_indexes, c))
        for i in range(s):
            C.memmove(C.byref(lift_params_list[i]), C.byref(arr[i]), C.sizeof(LiftParams))
        return l

    def zero_run_pack(self, coeffs, n, c, planar):
        """zero-run formation of the entropy loops -> (runs [m], values [m,c], trailing_run)"""
        co = np.ascontiguousarray(coeffs, dtype=np.int32).reshape(-1)
        runs = np.zeros(n, np.int32)
        vals = np.zeros(n * c, np.int32)
        m, tr = C.c_int32(), C.c_int32()
        _lib.check(self._lib.gpcc_zero_run_pack(self._h, co.ctypes.data, n, c, int(planar), runs.ctypes.data,
                                                vals.ctypes.data, C.byref(m), C.byref(tr)))
        return runs[:m.value].copy(), vals[:m.value * c].reshape(m.value, c).copy(), tr.value

    def raht_encode_attr_packed(self, params, xyz, attrs, bitdepth=8):
        """raht_encode_attr handing back the entropy loop's symbol stream ->
        (runs [m], values [m,c], trailing_run, clipped recon [n,c] point order)"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = a.shape
        runs = np.zeros(n, np.int32)
        vals = np.zeros(n * c, np.int32)
        m, tr = C.c_int32(), C.c_int32()
        _lib.check(self._lib.gpcc_raht_encode_attr_packed(self._h, C.byref(params), xyz.ctypes.data, a.ctypes.data,
                                                          runs.ctypes.data, vals.ctypes.data, C.byref(m),
                                                          C.byref(tr), n, c, bitdepth))
        return runs[:m.value].copy(), vals[:m.value * c].reshape(m.value, c).copy(), tr.value, a

    def raht_encode_attr_packed_regions(self, params, regions, xyz, attrs, bitdepth=8):
        """raht_encode_attr_packed with QP regions (params.qp_regions(...)): the per-point offsets of
        qpSet.regionQpOffset are derived on the device"""
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = a.shape
        runs = np.zeros(n, np.int32)
        vals = np.zeros(n * c, np.int32)
        m, tr = C.c_int32(), C.c_int32()
        _lib.check(self._lib.gpcc_raht_encode_attr_packed_regions(
            self._h, C.byref(params), C.byref(regions) if regions is not None else None, xyz.ctypes.data, a.ctypes.data,
            runs.ctypes.data, vals.ctypes.data, C.byref(m), C.byref(tr), n, c, bitdepth))
        return runs[:m.value].copy(), vals[:m.value * c].reshape(m.value, c).copy(), tr.value, a

    def raht_decode_attr_regions(self, params, regions, xyz, coeffs, c, bitdepth=8):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        n = xyz.shape[0]
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        a = np.zeros((n, c), dtype=np.int32)
        _lib.check(self._lib.gpcc_raht_decode_attr_regions(
            self._h, C.byref(params), C.byref(regions) if regions is not None else None, xyz.ctypes.data, a.ctypes.data,
            co.ctypes.data, n, c, bitdepth))
        return a

    def dev_pred_attr(self, encode, lod_params, pred_params_list, offsets, d_xyz, d_attrs, d_values, c, icp=None,
                      d_indexes=None):
        """gpcc_dev_pred_encode_attr / _decode_attr on device buffers; pred_params_list: one
        PredParams per slice (filled with the LoD structure); -> icp int8 [slices, 32, 3]"""
        from .params import PredParams
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        s = len(offs) - 1
        arr = (PredParams * s)(*pred_params_list)
        l = np.zeros((s, 32, 3), dtype=np.int8) if icp is None else np.ascontiguousarray(icp, dtype=np.int8).copy()
        f = self._lib.gpcc_dev_pred_encode_attr if encode else self._lib.gpcc_dev_pred_decode_attr
        _lib.check(f(self._h, C.byref(lod_params), C.cast(arr, C.c_void_p), s,
                     offs.ctypes.data_as(C.POINTER(C.c_int64)), d_xyz, d_attrs, d_values, l.ctypes.data, d_indexes, c))
        for i in range(s):
            C.memmove(C.byref(pred_params_list[i]), C.byref(arr[i]), C.sizeof(PredParams))
        return l

    def binarise_symbols(self, runs, values, trailing_run, c):
        """gpcc_binarise_symbols -> uint8 array of (context << 1 | bin) decisions"""
        runs = np.ascontiguousarray(runs, dtype=np.int32)
        vals = np.ascontiguousarray(values, dtype=np.int32).reshape(-1)
        m = len(runs)
        nb = C.c_int64()
        cap = 64 + 40 * (m + 1) * c
        out = np.zeros(cap, dtype=np.uint8)
        rc = self._lib.gpcc_binarise_symbols(self._h, runs.ctypes.data if m else None, vals.ctypes.data if m else None,
                                             m, int(trailing_run), c, out.ctypes.data, cap, C.byref(nb))
        if rc and nb.value > cap:  # very large magnitudes: once more with the exact size
            cap = nb.value
            out = np.zeros(cap, dtype=np.uint8)
            rc = self._lib.gpcc_binarise_symbols(self._h, runs.ctypes.data if m else None,
                                                 vals.ctypes.data if m else None, m, int(trailing_run), c,
                                                 out.ctypes.data, cap, C.byref(nb))
        _lib.check(rc)
        return out[:nb.value].copy()


class MultiContext:
    """gpcc_multi: one context per listed device, slices sharded across them, one
    gather (RCCL over xGMI when the devices are distinct) onto the first."""

    def __init__(self, devices):
        self._lib = _lib.load()
        h = C.c_void_p()
        arr = (C.c_int32 * len(devices))(*devices)
        _lib.check(self._lib.gpcc_multi_create(arr, len(devices), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpcc_multi_destroy(self._h)
            self._h = None

    def uses_rccl(self):
        return bool(self._lib.gpcc_multi_uses_rccl(self._h))

    def raht_forward(self, params, offsets, morton, attrs):
        """-> (coeffs planar per slice [c*N], recon [N, c])"""
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        rec = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = rec.shape
        co = np.zeros(c * n, dtype=np.int32)
        _lib.check(self._lib.gpcc_multi_raht_forward(
            self._h, C.byref(params), len(offs) - 1, offs.ctypes.data_as(C.POINTER(C.c_int64)),
            morton.ctypes.data, rec.ctypes.data, co.ctypes.data, c))
        return co, rec

    def raht_inverse(self, params, offsets, morton, coeffs, c):
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        rec = np.zeros((len(morton), c), dtype=np.int32)
        _lib.check(self._lib.gpcc_multi_raht_inverse(
            self._h, C.byref(params), len(offs) - 1, offs.ctypes.data_as(C.POINTER(C.c_int64)),
            morton.ctypes.data, rec.ctypes.data, co.ctypes.data, c))
        return rec

    def _lod_coder(self, name, lod_params, params_list, offsets, xyz, attrs, values, side, c):
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        s = len(offs) - 1
        blocks = (type(params_list[0]) * s)(*params_list)
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        idx = np.zeros(len(xyz), dtype=np.int32)
        _lib.check(getattr(self._lib, name)(
            self._h, C.byref(lod_params), C.cast(blocks, C.c_void_p), s, offs.ctypes.data_as(C.POINTER(C.c_int64)),
            xyz.ctypes.data, attrs.ctypes.data, values.ctypes.data, side.ctypes.data, idx.ctypes.data, c))
        for i in range(s):  # the LoD structure of every slice comes back in its block
            C.memmove(C.addressof(params_list[i]), C.addressof(blocks[i]), C.sizeof(blocks[i]))
        return idx

    def lod_encode_attr(self, predicting, lod_params, params_list, offsets, xyz, attrs):
        """gpcc_multi_lift_encode_attr / gpcc_multi_pred_encode_attr: every slice's LoD build +
        transform on the device its run belongs to -> (values [N,c] coding order per slice,
        recon [N,c], side int8 [slices,32] (lcp) or [slices,32,3] (icp), indexes [N])"""
        a = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = a.shape
        v = np.zeros((n, c), dtype=np.int32)
        side = np.zeros((len(offsets) - 1, 32, 3) if predicting else (len(offsets) - 1, 32), dtype=np.int8)
        name = "gpcc_multi_pred_encode_attr" if predicting else "gpcc_multi_lift_encode_attr"
        idx = self._lod_coder(name, lod_params, params_list, offsets, xyz, a, v, side, c)
        return v, a, side, idx

    def lod_decode_attr(self, predicting, lod_params, params_list, offsets, xyz, values, side):
        """gpcc_multi_lift_decode_attr / gpcc_multi_pred_decode_attr -> recon [N,c]"""
        v = np.ascontiguousarray(values, dtype=np.int32)
        n, c = v.shape
        a = np.zeros((n, c), dtype=np.int32)
        side = np.ascontiguousarray(side, dtype=np.int8)
        name = "gpcc_multi_pred_decode_attr" if predicting else "gpcc_multi_lift_decode_attr"
        self._lod_coder(name, lod_params, params_list, offsets, xyz, a, v, side, c)
        return a

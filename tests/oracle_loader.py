"""ctypes loaders for the two CPU checkers (TEST INFRASTRUCTURE):
  oracle/libgpcc_oracle.so      our plain-C restatement
  oracle/_ref/libtmc3_ref.so    the compiled reference (when present)
Both export the same call shapes (prefix oracle_ / ref_)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL)


class Checker:
    """Uniform wrapper over one checker library."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        for name in ("raht_forward", "raht_inverse"):
            f = getattr(self.lib, prefix + name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, _i64p, C.c_void_p, _i32p, _i32p, C.c_int32, C.c_int32]
        f = getattr(self.lib, prefix + "attr_morton_sort")
        f.restype = C.c_int
        f.argtypes = [_i32p, C.c_int32, _i64p, _i32p]

    def fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    def _qp(self, qp_off):
        if qp_off is None:
            return None, None
        q = np.ascontiguousarray(qp_off, dtype=np.int32)
        return q, q.ctypes.data_as(C.c_void_p)

    def raht_forward(self, params, morton, attrs, qp_off=None):
        """returns (coeffs [c*n] planar, recon [n,c])"""
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        rec = np.ascontiguousarray(attrs, dtype=np.int32).copy()
        n, c = rec.shape
        coeffs = np.zeros(c * n, dtype=np.int32)
        keep, qp = self._qp(qp_off)
        rc = getattr(self.lib, self.prefix + "raht_forward")(
            C.addressof(params), morton, qp, rec.reshape(-1), coeffs, n, c)
        assert rc == 0, rc
        return coeffs, rec

    def raht_inverse(self, params, morton, coeffs, c, qp_off=None):
        morton = np.ascontiguousarray(morton, dtype=np.int64)
        n = morton.shape[0]
        rec = np.zeros((n, c), dtype=np.int32)
        co = np.ascontiguousarray(coeffs, dtype=np.int32).copy()
        keep, qp = self._qp(qp_off)
        rc = getattr(self.lib, self.prefix + "raht_inverse")(
            C.addressof(params), morton, qp, rec.reshape(-1), co, n, c)
        assert rc == 0, rc
        return rec

    def recolour(self, params, src_xyz, src_attrs, tgt_xyz, scale=1.0, offset=(0, 0, 0)):
        """pcc::recolour: attributes of the source cloud transferred to tgt_xyz -> [nt, c]"""
        f = self.fn("recolour", C.c_int,
                    [C.c_void_p, _i32p, _i32p, C.c_int32, _i32p, C.c_int32, C.c_int32, C.c_float, _i32p, _i32p])
        sx = np.ascontiguousarray(src_xyz, dtype=np.int32)
        sa = np.ascontiguousarray(src_attrs, dtype=np.int32)
        tx = np.ascontiguousarray(tgt_xyz, dtype=np.int32)
        ns, c = sa.shape
        nt = tx.shape[0]
        out = np.zeros((nt, c), dtype=np.int32)
        off = np.ascontiguousarray(offset, dtype=np.int32)
        rc = f(C.addressof(params), sx.reshape(-1), sa.reshape(-1), ns, tx.reshape(-1), nt, c,
               float(scale), off, out.reshape(-1))
        assert rc == 0, rc
        return out

    def morton_sort(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        n = xyz.shape[0]
        morton = np.zeros(n, dtype=np.int64)
        order = np.zeros(n, dtype=np.int32)
        rc = getattr(self.lib, self.prefix + "attr_morton_sort")(xyz.reshape(-1), n, morton, order)
        assert rc == 0
        return morton, order


_cache = {}


def oracle():
    if "oracle" not in _cache:
        path = os.path.join(ORACLE_DIR, "libgpcc_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        _cache["oracle"] = Checker(path, "oracle_")
    return _cache["oracle"]


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libtmc3_ref.so"))


def ref():
    if "ref" not in _cache:
        _cache["ref"] = Checker(os.path.join(ORACLE_DIR, "_ref", "libtmc3_ref.so"), "ref_")
    return _cache["ref"]

"""ctypes loader of tests/emu/liblod_emu.so (TEST INFRASTRUCTURE): the scalable-lifting LoD build
of the library (lod_scalable.hpp + lod_kernels.hpp) compiled for the CPU wavefront emulator."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR, "liblod_emu.so"], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(os.path.join(EMU_DIR, "liblod_emu.so"))
        _lib.lod_emu_scalable_build.argtypes = [C.c_void_p, _i32p, C.c_int32, _i32p, _i32p, _i32p, _i32p, _i32p,
                                                C.POINTER(C.c_int32)]
        _lib.lod_emu_scalable_build.restype = C.c_int
        _lib.lod_emu_inter_build.argtypes = [C.c_void_p, _i32p, C.c_int32, _i32p, C.c_int32, C.c_int32, C.c_int32, _i32p,
                                             _i32p, _i32p, _i32p, _i32p, C.POINTER(C.c_int32), _i32p]
        _lib.lod_emu_inter_build.restype = C.c_int
        _lib.lift_emu_inter.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p,
                                        C.c_int32, _i32p]
        _lib.lift_emu_inter.restype = C.c_int
        _lib.pred_emu_inter.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p,
                                        C.c_int32, _i32p]
        _lib.pred_emu_inter.restype = C.c_int
    return _lib


def pred_inter(forward, pp, lod, attrs_ref, attrs=None, values=None):
    """the reflectance predicting transform with neighbours in a reference frame: the library's DAG
    pass (decoder; encoder without direct predictors) under the emulator -> (values [n,1], recon [n,1])"""
    n = len(lod["nc"])
    a = np.ascontiguousarray(attrs, dtype=np.int32).copy().reshape(-1) if forward else np.zeros(n, np.int32)
    v = np.zeros(n, np.int32) if forward else np.ascontiguousarray(values, dtype=np.int32).copy().reshape(-1)
    ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
    rc = lib().pred_emu_inter(C.addressof(pp), int(forward), n, np.ascontiguousarray(lod["nc"], dtype=np.int32),
                              np.ascontiguousarray(lod["ni"], dtype=np.int32).reshape(-1),
                              np.ascontiguousarray(np.asarray(lod["w"]).astype(np.int32)).reshape(-1),
                              np.ascontiguousarray(lod["ref"], dtype=np.int32).reshape(-1),
                              np.ascontiguousarray(lod["indexes"], dtype=np.int32), a, ar, len(ar), v)
    assert rc == 0, rc
    return v.reshape(n, 1), a.reshape(n, 1)


def lift_inter(forward, lf, lod, attrs_ref, attrs=None, coeffs=None):
    """reflectance lifting with neighbours in a reference frame, the library's arrangement under the
    emulator -> (coeffs [n,1], recon [n,1])"""
    n = len(lod["nc"])
    a = np.ascontiguousarray(attrs, dtype=np.int32).copy().reshape(-1) if forward else np.zeros(n, np.int32)
    co = np.zeros(n, np.int32) if forward else np.ascontiguousarray(coeffs, dtype=np.int32).copy().reshape(-1)
    ar = np.ascontiguousarray(attrs_ref, dtype=np.int32).reshape(-1)
    rc = lib().lift_emu_inter(C.addressof(lf), int(forward), n, np.ascontiguousarray(lod["nc"], dtype=np.int32),
                              np.ascontiguousarray(lod["ni"], dtype=np.int32).reshape(-1),
                              np.ascontiguousarray(np.asarray(lod["w"]).astype(np.int32)).reshape(-1),
                              np.ascontiguousarray(lod["ref"], dtype=np.int32).reshape(-1),
                              np.ascontiguousarray(lod["indexes"], dtype=np.int32), a, ar, len(ar), co)
    assert rc == 0, rc
    return co.reshape(n, 1), a.reshape(n, 1)


def inter_build(lp, xyz, xyz_ref, search_range, frame_distance=1):
    """the inter-prediction LoD build (periodic / centroid sub-sampling) under the emulator
    -> dict as lod_helpers.oracle_lod_generate_inter (weights int32)"""
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    xyz_ref = np.ascontiguousarray(xyz_ref, dtype=np.int32)
    n = len(xyz)
    nc = np.zeros(n, np.int32)
    ni = np.zeros((n, 3), np.int32)
    w = np.zeros((n, 3), np.int32)
    idx = np.zeros(n, np.int32)
    npl = np.zeros(32, np.int32)
    ref = np.zeros((n, 3), np.int32)
    nl = C.c_int32()
    rc = lib().lod_emu_inter_build(C.addressof(lp), xyz.reshape(-1), n, xyz_ref.reshape(-1), len(xyz_ref), search_range,
                                   frame_distance, nc, ni.reshape(-1), w.reshape(-1), idx, npl, C.byref(nl), ref.reshape(-1))
    assert rc == 0, rc
    return dict(nc=nc, ni=ni, w=w, indexes=idx, npl=npl[:nl.value].copy(), ref=ref)


def intra_build(lp, xyz):
    """the ordinary (intra, non-scalable) LoD build -- all three sub-samplers, the distance one with its
    workgroups running together -- under the emulator -> dict as lod_helpers.oracle_lod_generate"""
    o = inter_build(lp, xyz, np.zeros((0, 3), np.int32), 0, 0)
    del o["ref"]
    return o


def scalable_build(lp, xyz):
    """-> dict as lod_helpers.oracle_lod_generate (weights int32)"""
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    n = len(xyz)
    nc = np.zeros(n, np.int32)
    ni = np.zeros((n, 3), np.int32)
    w = np.zeros((n, 3), np.int32)
    idx = np.zeros(n, np.int32)
    npl = np.zeros(32, np.int32)
    nl = C.c_int32()
    rc = lib().lod_emu_scalable_build(C.addressof(lp), xyz.reshape(-1), n, nc, ni.reshape(-1), w.reshape(-1), idx, npl,
                                      C.byref(nl))
    assert rc == 0, rc
    return dict(nc=nc, ni=ni, w=w, indexes=idx, npl=npl[:nl.value].copy())


def assert_same_lod(got, want, msg=""):
    """LoD structures equal; weights compared for the neighbours that exist (under scalable lifting the
    reference leaves the raw squared distance of a pruned neighbour in its slot, nobody reads it)"""
    for k in ("npl", "indexes", "nc", "ni"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=f"{msg} {k}")
    live = np.arange(3)[None, :] < np.asarray(want["nc"])[:, None]
    np.testing.assert_array_equal(np.asarray(got["w"]).astype(np.uint64)[live], np.asarray(want["w"]).astype(np.uint64)[live],
                                  err_msg=f"{msg} w")

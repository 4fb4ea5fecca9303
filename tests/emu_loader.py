"""ctypes loader of tests/emu/libcx_emu.so (TEST INFRASTRUCTURE): the compact
level pass of the library compiled for the CPU wavefront emulator."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(os.path.join(EMU_DIR, "libcx_emu.so"))
        _lib.cx_emu_transform.argtypes = [C.c_void_p, C.c_int, C.c_int32, _i64p, _i64p, _i32p, _i32p,
                                          C.c_int32, C.c_int32, C.c_void_p]
        _lib.cx_emu_transform.restype = C.c_int
        _lib.cx_emu_check_tree.argtypes = [C.c_int32, _i64p, _i64p, C.c_int32]
        _lib.cx_emu_check_tree.restype = C.c_int
        _lib.cx_emu_check_links.argtypes = [C.c_int32, _i64p, _i64p, C.c_int32, C.c_int32]
        _lib.cx_emu_check_links.restype = C.c_int
        _lib.cx_emu_check_coeffs.argtypes = [C.c_int64, C.c_int64, C.c_uint64]
        _lib.cx_emu_check_coeffs.restype = C.c_int
        _lib.cx_emu_supported.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        _lib.cx_emu_supported.restype = C.c_int
    return _lib


def _bits(morton, offsets):
    b = 1
    for i in range(len(offsets) - 1):
        m = morton[offsets[i]:offsets[i + 1]]
        b = max(b, int(int(m[0]) ^ int(m[-1])).bit_length())
    return b


def supported(p, n, has_qp=False):
    return bool(lib().cx_emu_supported(C.addressof(p), int(has_qp), n))


def forward(p, morton, attrs, offsets=None, f64=False, links=False):
    """-> (coeffs planar per slice, recon [n, c]) from the emulated kernels (f64: the level
    kernels in ArithF64; an out-of-range value makes the harness return -103)"""
    n, c = attrs.shape
    offs = np.ascontiguousarray([0, n] if offsets is None else offsets, dtype=np.int64)
    morton = np.ascontiguousarray(morton, dtype=np.int64)
    rec = np.ascontiguousarray(attrs, dtype=np.int32).copy().reshape(-1)
    co = np.zeros(n * c, dtype=np.int32)
    rc = lib().cx_emu_transform(C.addressof(p), 1 | (2 if f64 else 0) | (4 if links else 0), len(offs) - 1, offs, morton, rec, co, c, _bits(morton, offs), None)
    assert rc == 0, rc
    return co, rec.reshape(n, c)


def inverse(p, morton, coeffs, c, offsets=None, f64=False, links=False):
    n = len(morton)
    offs = np.ascontiguousarray([0, n] if offsets is None else offsets, dtype=np.int64)
    morton = np.ascontiguousarray(morton, dtype=np.int64)
    rec = np.zeros(n * c, dtype=np.int32)
    co = np.ascontiguousarray(coeffs, dtype=np.int32).copy()
    rc = lib().cx_emu_transform(C.addressof(p), (2 if f64 else 0) | (4 if links else 0), len(offs) - 1, offs, morton, rec, co, c, _bits(morton, offs), None)
    assert rc == 0, rc
    return rec.reshape(n, c)


def check_links(morton, offsets, bits, use_top):
    offs = np.ascontiguousarray(offsets, dtype=np.int64)
    return lib().cx_emu_check_links(len(offs) - 1, offs, np.ascontiguousarray(morton, dtype=np.int64), bits, int(use_top))


def check_tree(morton, offsets, bits):
    offs = np.ascontiguousarray(offsets, dtype=np.int64)
    return lib().cx_emu_check_tree(len(offs) - 1, offs, np.ascontiguousarray(morton, dtype=np.int64), bits)

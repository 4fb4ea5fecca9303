"""ctypes loader of tests/emu/libsweep_emu.so (TEST INFRASTRUCTURE): the block loop with sub-node
prediction -- raht_sweep.hpp for a slice's coarse levels, raht_subnode.hpp for the others --
compiled for the CPU wavefront emulator."""
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
        subprocess.run(["make", "-s", "-C", EMU_DIR, "libsweep_emu.so"], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(os.path.join(EMU_DIR, "libsweep_emu.so"))
        _lib.sweep_emu_transform.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, _i64p, _i64p, _i32p, _i32p,
                                             C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        _lib.sweep_emu_transform.restype = C.c_int
    return _lib


def _bits(morton, offsets):
    b = 1
    for i in range(len(offsets) - 1):
        m = morton[offsets[i]:offsets[i + 1]]
        b = max(b, int(int(m[0]) ^ int(m[-1])).bit_length())
    return b


def _run(p, flags, sweep_parents, morton, rec, co, c, offsets):
    n = len(morton)
    offs = np.ascontiguousarray([0, n] if offsets is None else offsets, dtype=np.int64)
    morton = np.ascontiguousarray(morton, dtype=np.int64)
    swept = C.c_int32(0)
    rc = lib().sweep_emu_transform(C.addressof(p), flags, sweep_parents, len(offs) - 1, offs, morton, rec, co, c,
                                   _bits(morton, offs), C.byref(swept))
    assert rc == 0, rc
    return swept.value


def forward(p, morton, attrs, offsets=None, f64=False, sweep_parents=8192, rec=False):
    """-> (coeffs planar per slice, recon [n, c], levels taken by the sweep kernel)"""
    n, c = attrs.shape
    out = np.ascontiguousarray(attrs, dtype=np.int32).copy().reshape(-1)
    co = np.zeros(n * c, dtype=np.int32)
    swept = _run(p, 1 | (2 if f64 else 0) | (4 if rec else 0), sweep_parents, morton, out, co, c, offsets)
    return co, out.reshape(n, c), swept


def inverse(p, morton, coeffs, c, offsets=None, f64=False, sweep_parents=8192, rec=False):
    n = len(morton)
    out = np.zeros(n * c, dtype=np.int32)
    co = np.ascontiguousarray(coeffs, dtype=np.int32).copy()
    swept = _run(p, (2 if f64 else 0) | (4 if rec else 0), sweep_parents, morton, out, co, c, offsets)
    return out.reshape(n, c), swept

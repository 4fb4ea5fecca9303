"""CPU tier: gpcc_recolour's kernels (recolour_kdtree.hpp: nanoflann's k-d tree built level by
level and searched in nanoflann's order; recolour_kernels.hpp: forward / backward / lists in the
reference's order / blend) under the CPU wavefront emulator, against the oracle restatement and --
where present -- the compiled reference itself, ties included."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_loader as ol
from mpeg_pcc_tmc13_amd import recolour_params, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR, "librc_emu.so"], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(os.path.join(EMU_DIR, "librc_emu.so"))
        _lib.rc_emu_recolour.restype = C.c_int
        _lib.rc_emu_recolour.argtypes = [C.c_void_p, _i32p, _i32p, C.c_int32, _i32p, C.c_int32, C.c_int32, C.c_float,
                                         _i32p, _i32p]
    return _lib


def emu_recolour(p, xyz, attrs, tgt, scale=1.0, offset=(0, 0, 0)):
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    attrs = np.ascontiguousarray(attrs, dtype=np.int32)
    if attrs.ndim == 1:
        attrs = attrs.reshape(-1, 1)
    tgt = np.ascontiguousarray(tgt, dtype=np.int32)
    out = np.zeros((len(tgt), attrs.shape[1]), dtype=np.int32)
    rc = lib().rc_emu_recolour(C.addressof(p), xyz.reshape(-1), attrs.reshape(-1), len(xyz), tgt.reshape(-1), len(tgt),
                               attrs.shape[1], scale, np.asarray(offset, dtype=np.int32), out.reshape(-1))
    assert rc == 0, rc
    return out


def requantise(xyz, scale):
    return np.unique(np.rint(xyz.astype(np.float64) * scale).astype(np.int32), axis=0)


CASES = [("dense", 0.5, {}), ("dense", 0.25, dict(k_bwd=2)), ("dense", 1.0, {}), ("dense", 0.37, dict(max_attr_fwd=200.0)),
         ("lidar", 0.25, {}), ("lidar", 0.013, dict(k_fwd=3, skip_fwd=False)),
         ("dense", 0.125, dict(k_bwd=4, max_attr_bwd=400.0)),    # backward lists beyond 16 entries
         ("dense", 0.5, dict(weighted_fwd=False, weighted_bwd=False, skip_bwd=True, search_range=2)),
         # a finite forward geometry limit (round 5): the reference's shrunk result vectors from the first target beyond it
         ("dense", 0.37, dict(max_geom_fwd=3.0)), ("dense", 0.5, dict(max_geom_fwd=1.5, k_fwd=4)),
         ("lidar", 0.013, dict(max_geom_fwd=4.0, skip_fwd=False))]


@pytest.mark.parametrize("kind,scale,kw", CASES)
def test_emulated_kernels_match_the_checkers(kind, scale, kw):
    n = 6000
    xyz, a = synth.dense_cloud(n, seed=4, bits=7) if kind == "dense" else synth.lidar_cloud(n, seed=4)
    tgt = requantise(xyz, scale)
    p = recolour_params(bitdepth=8, **kw)
    got = emu_recolour(p, xyz, a, tgt, scale=scale)
    np.testing.assert_array_equal(got, ol.oracle().recolour(p, xyz, a, tgt, scale=scale))
    if ol.ref_available():
        np.testing.assert_array_equal(got, ol.ref().recolour(p, xyz, a, tgt, scale=scale))


def test_duplicates_offset_and_small_clouds():
    xyz, a = synth.dense_cloud(3000, seed=5, bits=6)
    xyz = np.concatenate([xyz, xyz[::3], xyz[::7]])
    a = np.concatenate([a, (a[::3] + 9) % 256, (a[::7] + 31) % 256]).astype(a.dtype)
    p = recolour_params(bitdepth=8)
    for scale, off in ((1.0, (0, 0, 0)), (0.5, (3, -2, 5))):
        tgt = np.unique(np.rint(xyz.astype(np.float64) * scale).astype(np.int32) - np.array(off, dtype=np.int32), axis=0)
        np.testing.assert_array_equal(emu_recolour(p, xyz, a, tgt, scale=scale, offset=off),
                                      ol.oracle().recolour(p, xyz, a, tgt, scale=scale, offset=off))
    xyz, a = synth.random_cloud(8, seed=2, bits=3, c=3)
    np.testing.assert_array_equal(emu_recolour(p, xyz, a, xyz), ol.oracle().recolour(p, xyz, a, xyz))
    np.testing.assert_array_equal(emu_recolour(p, xyz, a, xyz[:1]), ol.oracle().recolour(p, xyz, a, xyz[:1]))
    xyz, a = synth.random_cloud(700, seed=3, bits=5, c=1)
    tgt = requantise(xyz, 0.5)
    np.testing.assert_array_equal(emu_recolour(p, xyz, a, tgt, scale=0.5), ol.oracle().recolour(p, xyz, a, tgt, scale=0.5))


def test_several_levels_above_the_subtrees():
    """18 k points: four levels of the level-by-level build, then the wavefront-per-subtree kernel
    (<= 2 048 points each) -- a lidar sweep (uneven splits) and a dense cloud at a dyadic scale"""
    for kind, scale in (("lidar", 0.25), ("dense", 0.5)):
        xyz, a = synth.lidar_cloud(18000, seed=8) if kind == "lidar" else synth.dense_cloud(18000, seed=8, bits=8)
        tgt = requantise(xyz, scale)
        p = recolour_params(bitdepth=8)
        got = emu_recolour(p, xyz, a, tgt, scale=scale)
        want = ol.ref().recolour(p, xyz, a, tgt, scale=scale) if ol.ref_available() else ol.oracle().recolour(p, xyz, a, tgt, scale=scale)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2047, 2048, 2049, 70_001, 300_000, 2_200_000])
def test_tree_build_prefix_sum(n):
    """kd_scan (inclusive prefix sum, in place) with one block, a few, more than a wavefront's worth of blocks (147) and more
    than one block per thread of the middle kernel (1075 blocks): against numpy."""
    rng = np.random.default_rng(n)
    a = rng.integers(0, 3, n, dtype=np.int32)
    want = np.cumsum(a, dtype=np.int64).astype(np.int32)
    l = lib()
    l.rc_emu_scan.restype = C.c_int
    l.rc_emu_scan.argtypes = [_i32p, C.c_int64]
    assert l.rc_emu_scan(a, n) == 0
    np.testing.assert_array_equal(a, want)

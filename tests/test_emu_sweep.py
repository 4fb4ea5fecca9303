"""CPU tier: the coarse-level sweep of the block loop with sub-node prediction
(mpeg-pcc-tmc13_amd/csrc/raht_sweep.hpp: one workgroup per slice, the levels of a slice with few
parents inside ONE launch, hand-offs through LDS) under the lock-step wavefront emulator, against
the oracle.  The same kernel source the gfx950 library launches; the launch order of the library's
launch_transform is restated in tests/emu/sweep_emu_harness.cpp.  Pins the round / ring / zero-run
bookkeeping and the hand-over to the per-level kernels (raht_subnode.hpp) below the sweep; the
`-m gpu` tests remain the parity tests proper."""
import os

import numpy as np
import pytest

import emu_sweep_loader as es
import oracle_loader as ol
import raht_cases as rc
from mpeg_pcc_tmc13_amd import raht_params, synth


def _sub(case):
    pk = case["params"]
    return (pk.get("subnode", True) and not pk.get("haar", False) and pk.get("prediction", True)
            and case["qp_region"] is None and case["gen"][1].get("n", 0) <= 2000)


# the default CPU tier takes a cross-section (about three minutes under the emulator); GPCC_EMU_FULL=1: every case
FULL = os.environ.get("GPCC_EMU_FULL", "0") == "1"


def some(items, keep):
    """all items under GPCC_EMU_FULL=1, else those at the given positions"""
    items = list(items)
    return items if FULL else [items[i] for i in keep if i < len(items)]


ALL_ELIGIBLE = [c for c in rc.CASES if _sub(c)]
ELIGIBLE = some(ALL_ELIGIBLE, range(0, len(ALL_ELIGIBLE), 3))
F64 = pytest.mark.parametrize("f64", [False, True], ids=["i64", "f64"])


def test_case_table_has_eligible_cases():
    assert len(ALL_ELIGIBLE) >= 12 and len(ELIGIBLE) >= 4


def _check(p, morton, attrs, offsets=None, f64=False, sweep_parents=8192, min_swept=1, use_rec=False):
    c = attrs.shape[1]
    if offsets is None:
        o_co, o_rec = ol.oracle().raht_forward(p, morton, attrs)
    else:
        o_co = np.zeros(attrs.size, np.int32)
        o_rec = np.zeros_like(attrs)
        for a, b in zip(offsets[:-1], offsets[1:]):
            co, rec = ol.oracle().raht_forward(p, morton[a:b], attrs[a:b])
            o_co[a * c:b * c] = co
            o_rec[a:b] = rec
    co, out, swept = es.forward(p, morton, attrs, offsets=offsets, f64=f64, sweep_parents=sweep_parents, rec=use_rec)
    assert swept >= min_swept
    assert np.array_equal(co, o_co)
    assert np.array_equal(out, o_rec)
    inv, _ = es.inverse(p, morton, o_co, c, offsets=offsets, f64=f64, sweep_parents=sweep_parents, rec=use_rec)
    assert np.array_equal(inv, o_rec)


@pytest.mark.parametrize("case", ELIGIBLE, ids=[c["name"] for c in ELIGIBLE])
def test_sweep_matches_the_oracle(case):
    p, morton, attrs, _ = rc.make_inputs(case)
    if len(morton) < 2:
        pytest.skip("a single point has no level")
    _check(p, morton, attrs, f64=bool(p.raht_extension) and case["gen"][1].get("bitdepth", 8) <= 8,
           min_swept=1 if len(np.unique(morton)) > 1 else 0)


# the sweep takes the top levels and hands over to the per-level kernels where a level has more
# parents than `sweep_parents`: every split point of a small tree
@pytest.mark.parametrize("c,sweep_parents", some([(c, sp) for c in (1, 3) for sp in (1, 8, 64, 512)], [1, 6]))
def test_hand_over_to_the_level_kernels(c, sweep_parents):
    xyz, attrs = synth.random_cloud(n=2500 if FULL else 1500, seed=31 + c, bits=5, c=c, dup_fraction=0.1)
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    _check(raht_params(qp=28), morton, attrs, sweep_parents=sweep_parents, min_swept=1)


# the zero-run state (tmc3/RAHT.cpp:1618-1669) binds where many coefficients sit in the undecided
# band: every rate point, smooth and textured fields, with the rounds of EIGHT wavefronts in flight
@pytest.mark.parametrize("qp,f64", some([(q, f) for q in (4, 10, 16, 22, 28, 34, 40, 46, 51) for f in (False, True)], [3, 6, 11]))
def test_lossy_qp_sweep(qp, f64):
    xyz, a = synth.lidar_cloud(4000 if FULL else 2000, seed=qp, refl_noise=6 + (qp % 3) * 9)
    morton, attrs, _ = synth.sort_by_morton(xyz, a)
    _check(raht_params(qp=qp), morton, attrs, f64=f64)


@pytest.mark.parametrize("c", some([1, 2, 3], [2]))
def test_dense_surface(c):
    xyz, col = synth.dense_cloud(5000 if FULL else 2500, seed=9, bits=7)
    morton, attrs, _ = synth.sort_by_morton(xyz, col[:, :c])
    _check(raht_params(qp=34 - 6 * c), morton, attrs)


def test_batch_of_ragged_slices():
    parts = []
    for i, n in enumerate([1, 700, 2, 1500, 40, 9, 1200]):
        if i % 3 == 0:
            xyz, a = synth.lidar_cloud(max(n, 4), seed=40 + i)
        else:
            xyz, a = synth.random_cloud(n=n, seed=40 + i, bits=2 + i % 4, c=1, dup_fraction=0.2 if i % 2 else 0.0)
        m, a, _ = synth.sort_by_morton(xyz[:n], a[:n])
        parts.append((m, a))
    morton = np.concatenate([m for m, _ in parts])
    attrs = np.concatenate([a for _, a in parts])
    offs = np.concatenate([[0], np.cumsum([len(m) for m, _ in parts])])
    _check(raht_params(qp=22), morton, attrs, offsets=offs, min_swept=1)


VARIANTS = [dict(search_range=8), dict(threshold0=4, threshold1=10), dict(weights=(4, 2, 1, 3, 1)),
            dict(layers=[(30, -1), (34, -2), (38, 0), (28, 1)]),
            dict(ac_offsets=[[(i - 3, 3 - i) for i in range(7)], [(2, 1)] * 7, [(-4, 0)] * 7]),
            dict(qp=40, bitdepth=10), dict(extension=False)]


@pytest.mark.parametrize("vi", some(range(len(VARIANTS)), [0, 4, 6]))
def test_parameter_variants(vi):
    kw = dict(VARIANTS[vi])
    xyz, attrs = synth.random_cloud(n=1800 + 100 * vi, seed=60 + vi, bits=5, c=3 if vi % 2 else 1,
                                    dup_fraction=0.1 if vi % 2 else 0.0, bitdepth=kw.get("bitdepth", 8))
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    _check(raht_params(**kw), morton, attrs)


# the per-level kernels with the static half of a round taken from block records (raht_level_sub_kernel<.., REC>,
# GPCC_REC=1 in the library): every level below the sweep, and all levels (sweep_parents = 0)
@pytest.mark.parametrize("c,f64,ext,sweep_parents", some(
    [(c, f, e, sp) for sp in (0, 32) for c, f, e in ((1, True, True), (3, False, True), (1, False, False))], [0, 4, 5]))
def test_level_kernels_from_block_records(c, f64, ext, sweep_parents):
    xyz, attrs = synth.random_cloud(n=2600 if FULL else 1500, seed=71 + c, bits=5, c=c, dup_fraction=0.1)
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    _check(raht_params(qp=22 if c == 1 else 34, extension=ext), morton, attrs, f64=f64 and ext, sweep_parents=sweep_parents,
           min_swept=0, use_rec=True)


def test_level_kernels_from_block_records_ragged_batch():
    parts = []
    for i, n in enumerate([900, 3, 1400, 60]):
        xyz, a = synth.lidar_cloud(max(n, 4), seed=80 + i) if i % 2 == 0 else synth.random_cloud(n=n, seed=80 + i, bits=3, c=1)
        m, a, _ = synth.sort_by_morton(xyz[:n], a[:n])
        parts.append((m, a))
    morton = np.concatenate([m for m, _ in parts])
    attrs = np.concatenate([a for _, a in parts])
    offs = np.concatenate([[0], np.cumsum([len(m) for m, _ in parts])])
    _check(raht_params(qp=28), morton, attrs, offsets=offs, sweep_parents=16, min_swept=0, use_rec=True)

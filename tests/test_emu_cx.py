"""CPU tier: the compact level pass (mpeg-pcc-tmc13_amd/csrc/cx_*.hpp) run under the
lock-step wavefront emulator of tests/emu -- the same kernel sources and launch
sequence the gfx950 library uses, every thread a fiber -- against the oracle and the
committed golden vectors.  Pins the index arithmetic, the block lists, the in-pass
RDOQ hand-off and the one-irsqrt weight constants before a GPU is involved; the
`-m gpu` tests remain the parity tests proper (test_gpu_raht.py, test_gpu_tile.py)."""
import numpy as np
import pytest

import emu_loader as emu
import oracle_loader as ol
import raht_cases as rc
from mpeg_pcc_tmc13_amd import raht_params, synth

ELIGIBLE = [c for c in rc.CASES
            if c["qp_region"] is None and c["gen"][1].get("n", 0) <= 20000
            and emu.supported(rc.make_inputs(c)[0], 1)]


def test_case_table_has_eligible_cases():
    assert len(ELIGIBLE) >= 6


# both arithmetic back ends of the level kernels (raht_arith.hpp): int64 fixed point, and doubles
# where they are exact (what the library picks for attributes of at most 10 bits)
F64 = pytest.mark.parametrize("f64", [False, True], ids=["i64", "f64"])
# ... and with the opt-in neighbour links of raht_links.hpp (GPCC_LINKS=1) instead of the bisections
LINKS = pytest.mark.parametrize("links", [False, True], ids=["bisect", "links"])


@F64
@pytest.mark.parametrize("case", ELIGIBLE, ids=[c["name"] for c in ELIGIBLE])
def test_emulated_kernels_match_the_oracle(case, f64):
    p, morton, attrs, _ = rc.make_inputs(case)
    o_co, o_rec = ol.oracle().raht_forward(p, morton, attrs)
    co, rec = emu.forward(p, morton, attrs, f64=f64)
    assert np.array_equal(co, o_co)
    assert np.array_equal(rec, o_rec)
    assert np.array_equal(emu.inverse(p, morton, o_co, attrs.shape[1], f64=f64), o_rec)


VARIANTS = [dict(search_range=8), dict(threshold0=4, threshold1=10), dict(weights=(4, 2, 1, 3, 1)),
            dict(layers=[(30, -1), (34, -2), (38, 0), (28, 1)]),
            dict(ac_offsets=[[(i - 3, 3 - i) for i in range(7)], [(2, 1)] * 7, [(-4, 0)] * 7]),
            dict(qp=40, bitdepth=10), dict(qp=10), dict(prediction=False, qp=28)]


@LINKS
@F64
@pytest.mark.parametrize("vi", range(len(VARIANTS)))
@pytest.mark.parametrize("c", [1, 3])
def test_parameter_variants(vi, c, f64, links):
    if links and (f64 or c == 3 and vi not in (0, 1)):
        pytest.skip("the links are pinned on the int64 back end (the search is the same code in both)")
    kw = dict(VARIANTS[vi])
    kw.setdefault("subnode", False)
    xyz, attrs = synth.random_cloud(n=1500 + 100 * vi, seed=20 + vi, bits=5, c=c,
                                    dup_fraction=0.1 if vi % 2 else 0.0, bitdepth=kw.get("bitdepth", 8))
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    p = raht_params(**kw)
    o_co, o_rec = ol.oracle().raht_forward(p, morton, attrs)
    co, rec = emu.forward(p, morton, attrs, f64=f64, links=links)
    assert np.array_equal(co, o_co) and np.array_equal(rec, o_rec)
    assert np.array_equal(emu.inverse(p, morton, o_co, c, f64=f64, links=links), o_rec)


def _batch(sizes, c, seed):
    parts = []
    for i, n in enumerate(sizes):
        if c == 1 and i % 3 == 0:
            xyz, a = synth.lidar_cloud(int(n), seed=seed + i)
        else:
            xyz, a = synth.random_cloud(n=int(n), seed=seed + i, bits=2 + i % 4, c=c,
                                        dup_fraction=0.2 if i % 2 else 0.0)
        m, a, _ = synth.sort_by_morton(xyz[:n], a[:n])
        parts.append((m, a))
    morton = np.concatenate([m for m, _ in parts])
    attrs = np.concatenate([a for _, a in parts])
    offs = np.concatenate([[0], np.cumsum([len(m) for m, _ in parts])])
    return parts, morton, attrs, offs


@pytest.mark.parametrize("sizes,c", [([1, 2, 3, 700, 57, 64, 5000, 1, 333, 9, 2100], 1),
                                     (list(np.random.default_rng(5).integers(1, 40, size=150)), 3)],
                         ids=["ragged11", "tiny150"])
@LINKS
@F64
def test_ragged_batches(sizes, c, f64, links):
    if links and f64:
        pytest.skip("the links are pinned on the int64 back end")
    """slices of 1..5000 points in one batch (more than 64 slices: the plans are read
    from memory instead of LDS), every slice against the oracle"""
    p = raht_params(subnode=False, search_range=2500)
    parts, morton, attrs, offs = _batch(sizes, c, 30)
    co, rec = emu.forward(p, morton, attrs, offsets=offs, f64=f64, links=links)
    dec_in = np.zeros_like(co)
    for i, (m, a) in enumerate(parts):
        o_co, o_rec = ol.oracle().raht_forward(p, m, a)
        s0, s1 = int(offs[i]), int(offs[i + 1])
        assert np.array_equal(co[c * s0:c * s1], o_co), f"slice {i}"
        assert np.array_equal(rec[s0:s1], o_rec), f"slice {i}"
        dec_in[c * s0:c * s1] = o_co
    assert np.array_equal(emu.inverse(p, morton, dec_in, c, offsets=offs, f64=f64, links=links), rec)


@F64
@pytest.mark.parametrize("c", [1, 3])
def test_long_duplicate_chains_late_in_a_batch(c, f64):
    """slices of a handful of voxels with a hundred points each (finish_kernel's chains of
    (w, 1) transforms, weights beyond the small-weight tables) behind larger slices: the
    shape whose first chain coefficients came out wrong on the device in round 3
    (tests/test_gpu_batches.py) -- the logic, under the emulator"""
    sizes = [4000, 900, 3, 1072, 61, 658, 2]
    bits = [8, 6, 2, 1, 3, 1, 1]
    p = raht_params(qp=11, prediction=False, subnode=False)
    ms, as_ = [], []
    for i, (n, b) in enumerate(zip(sizes, bits)):
        xyz, a = synth.random_cloud(n, seed=510 + i, bits=b, c=c, dup_fraction=0.0)
        m, a, _ = synth.sort_by_morton(xyz, a)
        ms.append(m)
        as_.append(a)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    co, rec = emu.forward(p, np.concatenate(ms), np.concatenate(as_), offsets, f64=f64)
    inv = emu.inverse(p, np.concatenate(ms), co, c, offsets, f64=f64)
    o = ol.oracle()
    for i, n in enumerate(sizes):
        o_co, o_rec = o.raht_forward(p, ms[i], as_[i])
        b = int(offsets[i])
        assert np.array_equal(co[c * b:c * (b + n)], o_co), f"slice {i}"
        assert np.array_equal(rec[b:b + n], o_rec), f"slice {i}"
        assert np.array_equal(inv[b:b + n], o_rec), f"slice {i}"


def test_values_beyond_the_exact_range_are_reported():
    """ArithF64 with attributes far wider than the dispatcher admits (24 bits): the kernels' range
    check raises the sticky word (the harness returns -103) instead of handing out a different result;
    the same input in int64 is the oracle's"""
    rng = np.random.default_rng(3)
    xyz, _ = synth.random_cloud(n=3000, seed=77, bits=5, c=1)
    attrs = rng.integers(0, 1 << 24, size=(len(xyz), 1)).astype(np.int32)
    morton, attrs, _ = synth.sort_by_morton(xyz, attrs)
    p = raht_params(subnode=False, qp=40)
    with pytest.raises(AssertionError):
        emu.forward(p, morton, attrs, f64=True)
    o_co, o_rec = ol.oracle().raht_forward(p, morton, attrs)
    co, rec = emu.forward(p, morton, attrs)
    assert np.array_equal(co, o_co) and np.array_equal(rec, o_rec)


@pytest.mark.parametrize("n,bits,slices", [(1000, 12, 1), (5000, 30, 1), (3000, 9, 3), (70000, 36, 2),
                                           (2500, 54, 1), (5000, 20, 40), (4000, 10, 300)])
def test_tree_and_block_lists(n, bits, slices):
    """level arrays, head levels, value slots and the block lists of cx_tree.hpp against
    a direct computation (random codes incl. duplicates, several slices)"""
    rng = np.random.default_rng(n + bits)
    m = rng.integers(0, 1 << bits, size=n, dtype=np.int64)
    offs = [n * s // slices for s in range(slices + 1)]
    for s in range(slices):
        m[offs[s]:offs[s + 1]].sort()
    assert emu.check_tree(m, offs, bits) == 0


@pytest.mark.parametrize("n,bits,slices", [(700, 9, 1), (5000, 12, 1), (4000, 15, 3), (3000, 30, 2), (2500, 63, 1), (9, 6, 2),
                                           (6000, 18, 1)])
@pytest.mark.parametrize("use_top", [0, 1])
def test_neighbour_links(n, bits, slices, use_top):
    """raht_links.hpp: first children, occupancies and the 18 neighbour links of every node with more than one point,
    level by level, against a direct look-up of the keys inside the node's slice (random codes incl. duplicates,
    several slices; use_top: the driver's split between the single-workgroup loop and the per-level launches)"""
    rng = np.random.default_rng(7 * n + bits)
    m = rng.integers(0, 1 << bits, size=n, dtype=np.int64) if bits < 63 else rng.integers(0, (1 << 62) - 1, size=n, dtype=np.int64)
    if n > 100:  # clusters: chains of single-child nodes between the branchings, duplicates
        m[: n // 3] = (m[: n // 3] & ~np.int64(0x1FF)) | (m[: n // 3] & np.int64(7))
    offs = [n * s // slices for s in range(slices + 1)]
    for s in range(slices):
        m[offs[s]:offs[s + 1]].sort()
    assert emu.check_links(m, offs, bits, use_top) == 0


def test_weight_constants_from_one_irsqrt():
    """cx_norm / cx_coeffs (one irsqrt per weight, irsqrt(w << 30) == irsqrt(w) >> 15)
    against sqrt_weight / scale_rsqrt / raht_coeffs of raht_levels.hpp"""
    assert emu.lib().cx_emu_check_coeffs(70000, 300000, 12345) == 0

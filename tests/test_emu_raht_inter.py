"""CPU-only: the device kernels of RAHT with attribute inter prediction (csrc/raht_inter.hpp, raht_tile.hpp
with INTER, launch sequence raht_inter_driver.hpp) compiled for the CPU wavefront emulator (tests/emu) against
the oracle (oracle_raht_inter, itself pinned to the compiled reference by tests/test_oracle_raht_inter.py):
coefficients, reconstruction, decoder output, per-layer modes and filter taps, bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_loader as ol
from test_oracle_raht_inter import clouds, frame_of, region_offsets, run, run_qp

EMU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", EMU, "libinter_emu.so"], check=True)
    return C.CDLL(os.path.join(EMU, "libinter_emu.so"))


def check(lib, p, morton, attrs, mref, aref, depth, rdo, fest, skip, tag):
    o = ol.oracle().lib
    rc, co_o, rec_o, modes_o, taps_o = run(o, "oracle_raht_inter", p, True, morton, attrs, None, mref, aref, depth, rdo, fest, skip)
    assert rc == 0
    rc, co_e, rec_e, modes_e, taps_e = run(lib, "inter_emu_raht", p, True, morton, attrs, None, mref, aref, depth, rdo, fest, skip)
    assert rc == 0, (tag, rc)
    np.testing.assert_array_equal(taps_e, taps_o, err_msg=f"{tag} filter taps")
    np.testing.assert_array_equal(modes_e, modes_o, err_msg=f"{tag} layer modes")
    np.testing.assert_array_equal(co_e, co_o, err_msg=f"{tag} coefficients")
    np.testing.assert_array_equal(rec_e, rec_o, err_msg=f"{tag} encoder reconstruction")
    rc, _, dec_e, _, _ = run(lib, "inter_emu_raht", p, False, morton, attrs, co_o, mref, aref, depth, rdo, fest, skip, modes_o, taps_o)
    assert rc == 0, (tag, rc)
    np.testing.assert_array_equal(dec_e, rec_o, err_msg=f"{tag} decoder")
    return modes_o, taps_o


# GPCC_EMU_FULL=1: the clouds and frame / tool combinations of tests/test_gpu_raht_inter.py (40 CPU-minutes);
# the default tier runs the small clouds with a subset of the combinations
FULL = os.environ.get("GPCC_EMU_FULL") == "1"

VARIANTS = [dict(subnode=False), dict(prediction=False), dict(subnode=False, qp=22), dict(subnode=False, extension=False),
            dict(subnode=False, qp=46, chroma_offset=0),
            # the reference's default flag: sub-node prediction (the dependency kernels, two workspaces for the decision)
            dict(), dict(extension=False), dict(qp=22),
            # the integer Haar kernel (the lossless configurations): the frame's own level arrays
            dict(haar=True, qp=4, chroma_offset=0, subnode=False), dict(haar=True, qp=4, chroma_offset=0)]


@pytest.mark.parametrize("vi", range(len(VARIANTS)))
@pytest.mark.parametrize("rdo,fest", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_emulated_inter_raht(lib, vi, rdo, fest):
    from mpeg_pcc_tmc13_amd import raht_params, synth
    kw = VARIANTS[vi]
    rng = np.random.default_rng(3)
    seen_modes, seen_taps = set(), set()
    for name, xyz, attrs in clouds():
        if name == "one":
            continue
        if not FULL:
            if name in ("lidar", "dups") or (vi not in (0, 5) and (rdo, fest) != (1, 1)):
                continue
            xyz, attrs = xyz[:900], attrs[:900]
        morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
        for shift, jitter in ((0, 2), (0, 40), (40, 6)) if FULL else ((0, 2), (40, 6)):
            mref, aref = frame_of(xyz, attrs, rng, shift=shift, jitter=jitter)
            if kw.get("haar") and len(mref) > 1 and (int(mref[0] ^ mref[-1]).bit_length() - int(morton[0] ^ morton[-1]).bit_length()) % 3:
                # declined: the two trees do not line up on octree levels (raht_inter_driver.hpp, inter_supported)
                assert run(lib, "inter_emu_raht", raht_params(**kw), True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3)[0] == -2
                continue
            for depth, skip in ((15, 0), (2, 3), (15, 3)) if FULL else ((15, 3),):
                m, t = check(lib, raht_params(**kw), morton, a_sorted, mref, aref, depth, rdo, fest, skip,
                             f"{name} {kw} shift{shift} jitter{jitter} depth{depth} skip{skip} rdo{rdo} fest{fest}")
                seen_modes.update(m.tolist())
                seen_taps.update(t.tolist())
    if rdo and kw.get("prediction", True) and FULL:
        assert seen_modes == {0, 1}, seen_modes
    if fest and FULL:
        assert len(seen_taps) > 1, seen_taps


@pytest.mark.parametrize("kw", [dict(subnode=False), dict(), dict(haar=True, qp=4, chroma_offset=0), dict(subnode=False, extension=False)])
def test_emulated_inter_raht_with_region_qp_offsets(lib, kw):
    """per-point QP offsets of a region together with inter prediction: every kernel family (tile, dependency, Haar)"""
    from mpeg_pcc_tmc13_amd import raht_params, synth
    rng = np.random.default_rng(8)
    xyz, attrs = [c for c in clouds() if c[0] == "dense"][0][1:]
    xyz, attrs = xyz[:900], attrs[:900]
    morton, a_sorted, order = synth.sort_by_morton(xyz, attrs)
    q = region_offsets(xyz[order], rng)
    mref, aref = frame_of(xyz, attrs, rng, jitter=4)
    for rdo, fest in ((1, 1), (0, 0)):
        p = raht_params(**kw)
        rc, co_o, rec_o, modes_o, taps_o = run_qp(ol.oracle().lib, "oracle_raht_inter_qp", p, True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3, q)
        assert rc == 0
        rc, co_e, rec_e, modes_e, taps_e = run_qp(lib, "inter_emu_raht_qp", p, True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3, q)
        assert rc == 0
        np.testing.assert_array_equal(taps_e, taps_o)
        np.testing.assert_array_equal(modes_e, modes_o)
        np.testing.assert_array_equal(co_e, co_o)
        np.testing.assert_array_equal(rec_e, rec_o)
        rc, _, dec_e, _, _ = run_qp(lib, "inter_emu_raht_qp", p, False, morton, a_sorted, co_o, mref, aref, 15, rdo, fest, 3, q, modes_o, taps_o)
        assert rc == 0
        np.testing.assert_array_equal(dec_e, rec_o)


@pytest.mark.parametrize("kw", [dict(), dict(extension=False), dict(qp=16), dict(haar=True, qp=4, chroma_offset=0),
                                dict(subnode=False, extension=False), dict(subnode=False, haar=True, qp=4, chroma_offset=0)])
@pytest.mark.parametrize("links", [False, True], ids=["bisect", "links"])
def test_emulated_intra_level_kernels(lib, kw, links, monkeypatch):
    """the level kernels of the INTRA path under the emulator: with a depth limit of zero no level looks at the frame,
    and the driver runs the dependency kernels of the reference's default flags (raht_subnode.hpp: lossy, integer Haar,
    decoder) or the tile kernels exactly as gpcc_raht_forward / _inverse launch them -- against the plain intra oracle"""
    from mpeg_pcc_tmc13_amd import raht_params, synth
    # (links: the opt-in neighbour links of raht_links.hpp in the dependency kernels -- the library reads the switch
    # at every call)
    if links and (kw.get("extension") is False or kw.get("subnode") is False):
        pytest.skip("the links serve the dependency kernels with the RAHT extension only")
    monkeypatch.setenv("GPCC_LINKS", "1" if links else "0")
    o = ol.oracle()
    for name, xyz, attrs in clouds():
        if name in ("one", "lidar"):
            continue
        xyz, attrs = xyz[:1200], attrs[:1200]
        morton, a_sorted, _ = synth.sort_by_morton(xyz, attrs)
        p = raht_params(**kw)
        co_o, rec_o = o.raht_forward(p, morton, a_sorted)
        rc, co_e, rec_e, modes, taps = run(lib, "inter_emu_raht", p, True, morton, a_sorted, None, morton[:1], a_sorted[:1], -1, 1, 0, 0)
        assert rc == 0 and len(modes) == 0 and len(taps) == 0, (name, rc)
        np.testing.assert_array_equal(co_e, co_o, err_msg=f"{name} {kw} coefficients")
        np.testing.assert_array_equal(rec_e, rec_o, err_msg=f"{name} {kw} reconstruction")
        rc, _, dec_e, _, _ = run(lib, "inter_emu_raht", p, False, morton, a_sorted, co_o, morton[:1], a_sorted[:1], -1, 1, 0, 0)
        assert rc == 0
        np.testing.assert_array_equal(dec_e, rec_o, err_msg=f"{name} {kw} decoder")


@pytest.mark.parametrize("claim", [3, 8])
def test_emulated_claims_of_several_rounds(lib, claim, monkeypatch):
    """the opt-in claim form of the lossy sub-node encoder (raht_subnode.hpp, GPCC_SUB_CLAIM = R consecutive rounds per
    wavefront, the zero-run state carried between them in registers): measured slower on the MI355X
    (profiles/r05_claim_rounds_ab.txt) and therefore off, but it stays bit-exact -- noisy attributes at several qp put
    many coefficients into the RDOQ's undecided band, where the carried state decides"""
    from mpeg_pcc_tmc13_amd import raht_params, synth
    monkeypatch.setenv("GPCC_SUB_CLAIM", str(claim))
    o = ol.oracle()
    for seed, qp in (((1, 22), (3, 10)) if claim == 3 else ((2, 34),)):
        rng = np.random.default_rng(seed)
        xyz, a = synth.dense_cloud(1500, seed=seed, bits=6)
        a = np.clip(a[:, :1] + rng.integers(-20, 21, size=(len(a), 1)), 0, 255).astype(np.int32)
        morton, a_sorted, _ = synth.sort_by_morton(xyz, a)
        p = raht_params(qp=qp)
        co_o, rec_o = o.raht_forward(p, morton, a_sorted)
        rc, co_e, rec_e, _, _ = run(lib, "inter_emu_raht", p, True, morton, a_sorted, None, morton[:1], a_sorted[:1], -1, 1, 0, 0)
        assert rc == 0
        np.testing.assert_array_equal(co_e, co_o)
        np.testing.assert_array_equal(rec_e, rec_o)


def _seq_sum(t):
    s = 0.0
    for v in t.tolist():
        s += v
    return s


@pytest.mark.parametrize("case", ["costs", "ties", "ties_every_binade", "crossings", "zeros_tiny_huge", "negative",
                                  "short", "empty", "ragged"])
def test_rate_sum_is_the_sequential_sum(lib, case):
    """rate_sum_kernel replaces most of the chain of double additions by an integer sum per 512 terms (exact while
    the running sum stays in one binade and no term sits half-way between two grid points); everything else takes the
    chain.  Bit for bit against the additions done one after the other, on inputs built to hit every exit: ties (terms
    that are odd multiples of half the grid), binade crossings inside a chunk, a sum that starts at 0, terms too large
    for the grid, a negative term, chunk tails."""
    rng = np.random.default_rng(sum(map(ord, case)))
    n = 40_000
    if case == "costs":
        t = rng.uniform(0.0, 24.0, (2, n))
    elif case == "ties":
        # multiples of 2^-37: half-way cases whenever the grid is 2^-36 (sum in [2^16, 2^17)), exact otherwise
        t = rng.integers(0, 1 << 41, (2, n)).astype(np.float64) * 2.0 ** -37
    elif case == "ties_every_binade":
        # a term that is an odd multiple of half the CURRENT grid in every chunk, whatever the sum is by then
        t = rng.uniform(0.0, 8.0, (2, n))
        for e in range(2):
            s = 0.0
            for i in range(n):
                if i % 97 == 13 and s >= 1.0:
                    g = 2.0 ** (int(np.floor(np.log2(s))) - 52)
                    t[e, i] = (2 * int(rng.integers(1, 1 << 20)) + 1) * (g / 2)
                s += t[e, i]
    elif case == "crossings":
        t = rng.uniform(0.0, 4.0, (2, n))
        t[:, ::50] = 2.0 ** rng.integers(0, 30, t[:, ::50].shape)
    elif case == "zeros_tiny_huge":
        t = np.zeros((2, n))
        t[:, 5::7] = 1e-300
        t[:, 11::13] = rng.uniform(0, 3, t[:, 11::13].shape)
        t[0, 20_000] = 1e70
        t[1, 700] = 2.0 ** 60
    elif case == "negative":
        t = rng.uniform(0.0, 24.0, (2, n))
        t[0, 12_345] = -3.25
        t[1, 100] = -1e-9
    elif case == "short":
        t = rng.uniform(0.0, 24.0, (2, 300))
    elif case == "empty":
        t = np.zeros((2, 0))
    else:
        t = rng.uniform(0.0, 24.0, (2, 512 * 9 + 1))
    t = np.ascontiguousarray(t, dtype=np.float64)
    out = np.zeros(2)
    fn = lib.inter_emu_rate_sum
    fn.restype = C.c_int
    fn.argtypes = [np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_int32,
                   np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    assert fn(t.reshape(-1), t.shape[1], out) == 0
    want = np.array([_seq_sum(t[0]), _seq_sum(t[1])])
    assert out.tobytes() == want.tobytes(), (case, out, want, out - want)

"""rate_sum_kernel on the MI355X: the inter encoder's chain of double additions, most of it replaced by an integer sum
per 512 terms (csrc/raht_inter.hpp).  Bit for bit against the additions done one after the other, on inputs built to
hit every exit of the fast path -- the same cases as tests/test_emu_raht_inter.py runs under the emulator, here with
the hardware's own rounding, conversion and LDS chain."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _seq_sum(t):
    s = 0.0
    for v in t.tolist():
        s += v
    return s


def _terms(case, n=60_000):
    rng = np.random.default_rng(sum(map(ord, case)))
    if case == "costs":
        return rng.uniform(0.0, 24.0, (2, n))
    if case == "ties":
        return rng.integers(0, 1 << 41, (2, n)).astype(np.float64) * 2.0 ** -37
    if case == "ties_every_binade":
        t = rng.uniform(0.0, 8.0, (2, n))
        for e in range(2):
            s = 0.0
            for i in range(n):
                if i % 97 == 13 and s >= 1.0:
                    g = 2.0 ** (int(np.floor(np.log2(s))) - 52)
                    t[e, i] = (2 * int(rng.integers(1, 1 << 20)) + 1) * (g / 2)
                s += t[e, i]
        return t
    if case == "crossings":
        t = rng.uniform(0.0, 4.0, (2, n))
        t[:, ::50] = 2.0 ** rng.integers(0, 30, t[:, ::50].shape)
        return t
    if case == "zeros_tiny_huge":
        t = np.zeros((2, n))
        t[:, 5::7] = 1e-300
        t[:, 11::13] = rng.uniform(0, 3, t[:, 11::13].shape)
        t[0, 20_000] = 1e70
        t[1, 700] = 2.0 ** 60
        return t
    if case == "negative":
        t = rng.uniform(0.0, 24.0, (2, n))
        t[0, 12_345] = -3.25
        t[1, 100] = -1e-9
        return t
    if case == "short":
        return rng.uniform(0.0, 24.0, (2, 300))
    if case == "empty":
        return np.zeros((2, 0))
    if case == "level_sized":
        return rng.uniform(0.0, 24.0, (2, 1_000_003))
    return rng.uniform(0.0, 24.0, (2, 512 * 9 + 1))


@pytest.mark.parametrize("case", ["costs", "ties", "ties_every_binade", "crossings", "zeros_tiny_huge", "negative",
                                  "short", "empty", "ragged", "level_sized"])
def test_rate_sum_is_the_sequential_sum(case):
    from mpeg_pcc_tmc13_amd import _lib, context
    lib = _lib.load()
    ctx = context(0)
    t = np.ascontiguousarray(_terms(case), dtype=np.float64)
    out = np.zeros(2)
    fn = lib.gpcc_debug_rate_sum
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_int32,
                   np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    assert fn(ctx._h, t.reshape(-1), t.shape[1], out) == 0
    want = np.array([_seq_sum(t[0]), _seq_sum(t[1])])
    assert out.tobytes() == want.tobytes(), (case, out, want, out - want)

"""GPU parity of the Morton prologue: mortonAddr + the (code, index) order of
std::sort(MortonCodeWithIndex) (reference tmc3/AttributeEncoder.cpp:1316-1321,
PCCTMC3Common.h:184-190) against the CPU oracle.  Bit-exact, ties included."""
import numpy as np
import pytest

import oracle_loader as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    c = context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,bits", [(1, 3), (2, 1), (65, 2), (4096, 4), (4097, 10), (100000, 6),
                                    (300000, 18), (1000, 21)])
def test_sort_matches_oracle(n, bits, ctx):
    rng = np.random.default_rng(n * 31 + bits)
    xyz = rng.integers(0, 1 << bits, size=(n, 3)).astype(np.int32)
    m, o = ctx.morton_sort(xyz)
    om, oo = ol.oracle().morton_sort(xyz)
    np.testing.assert_array_equal(m, om)
    np.testing.assert_array_equal(o, oo)


def test_sort_many_duplicates_is_stable(ctx):
    xyz = np.zeros((50000, 3), dtype=np.int32)
    xyz[::3] = 1
    m, o = ctx.morton_sort(xyz)
    assert np.all(np.diff(m) >= 0)
    same = np.diff(m) == 0
    assert np.all(np.diff(o)[same] > 0)


def test_sort_rejects_out_of_range(ctx):
    from mpeg_pcc_tmc13_amd._lib import GpccError
    with pytest.raises(GpccError):
        ctx.morton_sort(np.array([[0, 0, 1 << 21]], dtype=np.int32))
    with pytest.raises(GpccError):
        ctx.morton_sort(np.array([[0, -1, 0]], dtype=np.int32))


def test_batched_sort_device_tier(ctx):
    import ctypes as C
    import torch
    from mpeg_pcc_tmc13_amd import _lib
    rng = np.random.default_rng(9)
    sizes = [5000, 1, 12345, 4096]
    xs = [rng.integers(0, 1 << 7, size=(n, 3)).astype(np.int32) for n in sizes]
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    dev = torch.device("cuda:0")
    d_x = torch.from_numpy(np.concatenate(xs).reshape(-1)).to(dev)
    d_m = torch.zeros(int(offsets[-1]), dtype=torch.int64, device=dev)
    d_o = torch.zeros(int(offsets[-1]), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    off = (C.c_int64 * len(offsets))(*[int(v) for v in offsets])
    ctx.set_morton_bits(21)
    _lib.check(_lib.load().gpcc_dev_attr_morton_sort(
        ctx._h, len(sizes), off, C.c_void_p(d_x.data_ptr()), C.c_void_p(d_m.data_ptr()),
        C.c_void_p(d_o.data_ptr())))
    ctx.synchronize()
    ctx.set_morton_bits(0)
    m, o = d_m.cpu().numpy(), d_o.cpu().numpy()
    for i, n in enumerate(sizes):
        om, oo = ol.oracle().morton_sort(xs[i])
        b = int(offsets[i])
        np.testing.assert_array_equal(m[b:b + n], om)
        np.testing.assert_array_equal(o[b:b + n], oo)

"""CPU-only: the C-ABI library loads and exports every symbol the header
declares; no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gpcc_attr_mi355.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpcc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from mpeg_pcc_tmc13_amd import build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from mpeg_pcc_tmc13_amd import _lib
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == names


def test_abi_version_and_struct_size(lib):
    from mpeg_pcc_tmc13_amd import RahtParams
    from mpeg_pcc_tmc13_amd import _lib
    import re
    hdr = open(os.path.join(ROOT, "include", "gpcc_attr_mi355.h")).read()
    assert int(re.search(r"#define GPCC_ABI_VERSION (\d+)", hdr).group(1)) == 6
    assert lib.gpcc_abi_version() == 6 == _lib.ABI_VERSION
    # 6 + 19 + 12 + 1 + 1 + 64 + 1 + 1 + 1 + 448 ints
    assert C.sizeof(RahtParams) == 4 * (6 + 19 + 12 + 1 + 1 + 64 + 3 + 32 * 7 * 2)


def test_prediction_weights_match_python_mirror(lib):
    from mpeg_pcc_tmc13_amd import RahtParams, raht_params
    p = RahtParams()
    w = (C.c_int32 * 5)(9, 3, 1, 5, 2)
    lib.gpcc_raht_set_prediction_weights(C.byref(p), w)
    q = raht_params()
    assert list(p.pred_weight_parent) == list(q.pred_weight_parent)
    assert list(p.pred_weight_child) == list(q.pred_weight_child)


def test_no_device_fails_loudly(lib):
    """Without a GPU the context cannot be created: an error code and a
    message, never a silent CPU path."""
    if lib.gpcc_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.gpcc_ctx_create(0, None, C.byref(h))
    assert rc == -3 and not h.value
    assert b"device" in lib.gpcc_last_error().lower()


def test_null_context_is_rejected_everywhere(lib):
    """Every compute entry refuses a null context with GPCC_ERR_INVALID_ARG
    (and says why) before touching any buffer -- CPU box included."""
    from mpeg_pcc_tmc13_amd import LiftParams, LodParams, RahtParams
    from mpeg_pcc_tmc13_amd import PredParams
    rp, lp, lf, pp = RahtParams(), LodParams(), LiftParams(), PredParams()
    dummy = C.cast((C.c_int32 * 3)(), C.c_void_p)   # a non-null inter_ref: the context is what is refused
    z = None
    off = (C.c_int64 * 2)(0, 1)
    out = C.c_int32()
    calls = [
        lambda: lib.gpcc_raht_forward(z, C.byref(rp), z, z, z, z, 1, 1),
        lambda: lib.gpcc_raht_inverse(z, C.byref(rp), z, z, z, z, 1, 1),
        lambda: lib.gpcc_attr_morton_sort(z, z, 1, z, z),
        lambda: lib.gpcc_dev_raht_forward(z, C.byref(rp), 1, off, z, z, z, z, 1),
        lambda: lib.gpcc_dev_raht_inverse(z, C.byref(rp), 1, off, z, z, z, z, 1),
        lambda: lib.gpcc_dev_attr_morton_sort(z, 1, off, z, z, z),
        lambda: lib.gpcc_lift_forward(z, C.byref(lf), 1, 1, z, z, z, z, z, z, z, z),
        lambda: lib.gpcc_lift_inverse(z, C.byref(lf), 1, 1, z, z, z, z, z, z, z, z),
        lambda: lib.gpcc_lod_compute_weights(z, 1, z, z, z),
        lambda: lib.gpcc_lod_build(z, C.byref(lp), z, 1, z, z, z, z, z, C.byref(out)),
        lambda: lib.gpcc_estimate_dist2(z, z, 1, 100, 128, C.c_float(0.85), C.byref(out)),
        lambda: lib.gpcc_raht_encode_attr(z, C.byref(rp), z, z, z, 1, 1, 8),
        lambda: lib.gpcc_raht_decode_attr(z, C.byref(rp), z, z, z, 1, 1, 8),
        lambda: lib.gpcc_lift_encode_attr(z, C.byref(lp), C.byref(lf), z, z, z, z, z, 1, 1),
        lambda: lib.gpcc_lift_decode_attr(z, C.byref(lp), C.byref(lf), z, z, z, z, z, 1, 1),
        lambda: lib.gpcc_zero_run_pack(z, z, 1, 1, 1, z, z, C.byref(out), C.byref(out)),
        lambda: lib.gpcc_raht_encode_attr_packed(z, C.byref(rp), z, z, C.byref(out), z, C.byref(out), C.byref(out), 1, 1, 8),
        # attribute inter prediction (round 3)
        lambda: lib.gpcc_lod_build_inter(z, C.byref(lp), z, 1, z, 1, 128, 1, z, z, z, z, z, C.byref(out), z),
        lambda: lib.gpcc_lift_forward_inter(z, C.byref(lf), 1, z, z, z, dummy, z, z, z, 1, z),
        lambda: lib.gpcc_lift_inverse_inter(z, C.byref(lf), 1, z, z, z, dummy, z, z, z, 1, z),
        lambda: lib.gpcc_pred_forward_inter(z, C.byref(pp), 1, z, z, z, dummy, z, z, z, 1, z),
        lambda: lib.gpcc_pred_inverse_inter(z, C.byref(pp), 1, z, z, z, dummy, z, z, z, 1, z),
        # RAHT with attribute inter prediction (round 4)
        lambda: lib.gpcc_raht_forward_inter(z, C.byref(rp), z, z, z, z, z, 1, 1, z, z, 1, z, C.byref(out), z, C.byref(out)),
        lambda: lib.gpcc_raht_inverse_inter(z, C.byref(rp), z, z, z, z, z, 1, 1, z, z, 1, z, 0, z, 0),
    ]
    for i, f in enumerate(calls):
        assert f() == -1, f"entry {i}"
        assert lib.gpcc_last_error()

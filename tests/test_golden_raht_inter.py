"""RAHT with attribute inter prediction against the committed record of the COMPILED REFERENCE
(tests/golden/raht_inter_golden.npz, made by tests/golden/make_raht_inter_golden.py): needs neither /root/reference nor
oracle/_ref.  The oracle (CPU tier), the kernels under the wavefront emulator (CPU tier) and the device (GPU tier)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle_loader as ol
import raht_inter_cases as rc
from test_oracle_raht_inter import run_qp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = np.load(os.path.join(HERE, "golden", "raht_inter_golden.npz"))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def inputs_of(case):
    p, morton, attrs, mref, aref, q = rc.make_inputs(case)
    assert str(GOLDEN[case[0] + "/in_sha"]) == sha(morton, attrs, mref, aref) + ("" if q is None else sha(q)), \
        "the generator's inputs changed: regenerate the fixture"
    return p, morton, attrs, mref, aref, q


def compare(name, co, rec, modes, taps, dec):
    np.testing.assert_array_equal(taps, GOLDEN[name + "/taps"], err_msg=f"{name} filter taps")
    np.testing.assert_array_equal(modes, GOLDEN[name + "/modes"], err_msg=f"{name} layer modes")
    np.testing.assert_array_equal(co, GOLDEN[name + "/coeffs"], err_msg=f"{name} coefficients")
    np.testing.assert_array_equal(rec, GOLDEN[name + "/rec"], err_msg=f"{name} encoder reconstruction")
    np.testing.assert_array_equal(dec, GOLDEN[name + "/rec"], err_msg=f"{name} decoder")


def through(lib, fn, case):
    name, _, _, _, _, depth, rdo, fest, skip, _ = case
    p, morton, attrs, mref, aref, q = inputs_of(case)
    rc_, co, rec, modes, taps = run_qp(lib, fn, p, True, morton, attrs, None, mref, aref, depth, rdo, fest, skip, q)
    assert rc_ == 0, (name, rc_)
    g = GOLDEN
    rc_, _, dec, _, _ = run_qp(lib, fn, p, False, morton, attrs, g[name + "/coeffs"], mref, aref, depth, rdo, fest, skip, q,
                               g[name + "/modes"], g[name + "/taps"])
    assert rc_ == 0, (name, rc_)
    compare(name, co, rec, modes, taps, dec)


@pytest.mark.parametrize("case", rc.CASES, ids=[c[0] for c in rc.CASES])
def test_oracle_equals_the_reference_record(case):
    through(ol.oracle().lib, "oracle_raht_inter_qp", case)


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(HERE, "emu")
    subprocess.run(["make", "-s", "-C", d, "libinter_emu.so"], check=True)
    return C.CDLL(os.path.join(d, "libinter_emu.so"))


@pytest.mark.parametrize("case", rc.CASES[::6], ids=[c[0] for c in rc.CASES[::6]])
def test_emulated_kernels_equal_the_reference_record(emu, case):
    through(emu, "inter_emu_raht_qp", case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", rc.CASES, ids=[c[0] for c in rc.CASES])
def test_device_equals_the_reference_record(case):
    from mpeg_pcc_tmc13_amd import RahtInterParams, context
    name, _, _, _, _, depth, rdo, fest, skip, _ = case
    p, morton, attrs, mref, aref, q = inputs_of(case)
    ctx = context(0)
    ip = RahtInterParams(depth, rdo, fest, skip)
    co, rec, modes, taps = ctx.raht_forward_inter(p, ip, morton, attrs, mref, aref, qp_off=q)
    dec = ctx.raht_inverse_inter(p, ip, morton, GOLDEN[name + "/coeffs"], attrs.shape[1], mref, aref, GOLDEN[name + "/modes"],
                                 GOLDEN[name + "/taps"], qp_off=q)
    compare(name, co, rec, modes, taps, dec)

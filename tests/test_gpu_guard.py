"""Guard mode of the library (GPCC_GUARD=1, csrc/gpcc_attr_mi355.hip): canary bands around every device allocation of
the library and behind every sub-allocation of a context's arena.  Round 4's GPU tier aborted once at a download
without a message -- exactly what an out-of-bounds device write into a neighbouring allocation
looks like -- so the tier can now be run with the bands armed (profiles/r05_gpu_tier_guarded.txt).  Here: the bands
catch a deliberate overflow (child processes, the parent's environment is not touched), and a normal transform under
guard mode is clean and has compared bands."""
import ctypes as C
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
g.load_package()
from mpeg_pcc_tmc13_amd import _lib, context, raht_params, synth
import numpy as np
lib = _lib.load()
ctx = context(0)
mode = int(sys.argv[1])
if mode < 2:
    lib.gpcc_debug_guard_selftest.argtypes = [C.c_void_p, C.c_int]
    rc = lib.gpcc_debug_guard_selftest(ctx._h, mode)
    lib.gpcc_ctx_synchronize(ctx._h)
    print("survived", rc)
else:
    lib.gpcc_debug_guard_checks.restype = C.c_ulonglong
    xyz, col = synth.dense_cloud(30000, seed=3, bits=9)
    m, a, _ = synth.sort_by_morton(xyz, col)
    for p in (raht_params(qp=34), raht_params(qp=34, subnode=False)):
        co, rec = ctx.raht_forward(p, m, a)
        inv = ctx.raht_inverse(p, m, co, a.shape[1])
        assert np.array_equal(inv, rec)
    lib.gpcc_ctx_synchronize(ctx._h)
    print("checks", lib.gpcc_debug_guard_checks())
"""


def run_child(mode, guard):
    env = dict(os.environ)
    env["GPCC_GUARD"] = "1" if guard else "0"
    return subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(mode)], env=env, capture_output=True,
                          text=True, timeout=600)


@pytest.mark.parametrize("mode", [0, 1])
def test_overflow_is_caught(mode):
    r = run_child(mode, guard=True)
    assert r.returncode != 0, r.stdout + r.stderr
    assert "GUARD BAND OVERWRITTEN" in r.stderr, r.stderr
    assert ("pool block" if mode == 0 else "arena") in r.stderr


def test_guards_are_off_by_default():
    r = run_child(0, guard=False)
    assert r.returncode == 0 and "survived 0" in r.stdout, r.stdout + r.stderr


def test_clean_run_compares_bands():
    r = run_child(2, guard=True)
    assert r.returncode == 0, r.stdout + r.stderr
    n = int(r.stdout.strip().split()[-1])
    assert n > 50, r.stdout

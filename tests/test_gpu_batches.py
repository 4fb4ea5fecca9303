"""Ragged batches through the device tier against the compiled reference: the batches of
tests/stress/stress_cx_batch.py that exposed the duplicate-chain failure of the shared finish kernel
(round 3: wrong first coefficients of long chains of duplicate points, C = 1, batches of 140 k+ points,
~70 % of the runs of such a batch), each run three times, and a short random sweep of both stress modes."""
import os
import sys

import pytest

import oracle_loader as ol

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ol.ref_available(), reason="compiled reference absent")]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress"))


@pytest.fixture(scope="module")
def ctx():
    from mpeg_pcc_tmc13_amd import context
    return context(0)


@pytest.mark.parametrize("it", [46, 144, 259, 404, 449, 483])
def test_batches_that_failed_in_round_3(it, ctx):
    import stress_cx_batch as sb
    p, c, ms, as_ = sb.make_batch(424200, it)
    o = ol.ref()
    want = [o.raht_forward(p, ms[i], as_[i]) for i in range(len(ms))]
    for rep in range(3):
        assert sb.run_batch(ctx, o, p, c, ms, as_, want) == [], f"batch {it}, run {rep}"


@pytest.mark.parametrize("allflags", [False, True])
def test_random_batches(allflags, ctx):
    import stress_cx_batch as sb
    o = ol.ref()
    for it in range(40):
        p, c, ms, as_ = sb.make_batch(990000 + (1000 if allflags else 0), it, allflags)
        assert sb.run_batch(ctx, o, p, c, ms, as_) == [], f"batch {it}"

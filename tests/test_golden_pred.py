"""The predicting transform against the committed record of the COMPILED REFERENCE's operator
(tests/golden/pred_golden.npz, made by tests/golden/make_pred_golden.py: symbols read back from the payload,
reconstruction, inter-component coefficients): needs neither /root/reference nor oracle/_ref.  The oracle builds its
OWN LoD structure here, so the record pins the chain LoD build -> mode decision with the running rate model ->
symbols.  CPU tier (the device is compared with the oracle by tests/test_gpu_pred.py)."""
import hashlib
import os

import numpy as np
import pytest

import lod_helpers as lh
import test_oracle_pred as tp
from golden.make_pred_golden import NAMES

GOLDEN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pred_golden.npz"))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def inputs_of(name):
    from mpeg_pcc_tmc13_amd import pred_params
    xyz, attrs, lp, qp, bitdepth, thr, po = tp.make(name)
    assert str(GOLDEN[name + "/in_sha"]) == sha(xyz, attrs), "the generator's inputs changed: regenerate the fixture"
    lod = lh.oracle_lod_generate(xyz, lp)
    pp = pred_params(lod["npl"], qp=qp, chroma_offset=0, bitdepth=bitdepth, threshold=thr,
                     max_levels=lp.num_detail_levels_minus1 + 1, **po)
    return xyz, attrs, lod, pp


@pytest.mark.parametrize("name", NAMES)
def test_oracle_encoder_equals_the_reference_record(name):
    xyz, attrs, lod, pp = inputs_of(name)
    values, rec, icp, _ = lh.oracle_pred(True, pp, lod, attrs=attrs)
    if attrs.shape[1] == 3 and pp.inter_component_prediction_enabled_flag:
        np.testing.assert_array_equal(icp, GOLDEN[name + "/icp"])
    np.testing.assert_array_equal(values, GOLDEN[name + "/values"])
    np.testing.assert_array_equal(rec, GOLDEN[name + "/rec"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_decoder_equals_the_reference_record(name):
    xyz, attrs, lod, pp = inputs_of(name)
    _, rec, _, _ = lh.oracle_pred(False, pp, lod, values=GOLDEN[name + "/values"], icp=GOLDEN[name + "/icp"])
    np.testing.assert_array_equal(rec, GOLDEN[name + "/rec"])

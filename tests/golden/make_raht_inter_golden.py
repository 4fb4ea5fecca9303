#!/usr/bin/env python3
"""Generate tests/golden/raht_inter_golden.npz from the COMPILED REFERENCE (oracle/_ref/libtmc3_ref.so, built by
oracle/Makefile from the sources under /root/reference): for every case of tests/raht_inter_cases.py the reference's
regionAdaptiveHierarchicalTransform / ...InverseTransform with attribute inter prediction -- coefficients, encoder
reconstruction, decoder output, attr_layer_code_mode, FilterTaps.  Inputs are not stored: they are regenerated from
the seeds (their SHA-256 is).  Run in the build container:

    make -C oracle && python tests/golden/make_raht_inter_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402,F401  (registers the package alias)
import oracle_loader as ol  # noqa: E402
import raht_inter_cases as rc  # noqa: E402
from test_oracle_raht_inter import run_qp  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ref = ol.ref().lib
    out = {}
    for case in rc.CASES:
        name, _, _, _, _, depth, rdo, fest, skip, _ = case
        p, morton, attrs, mref, aref, q = rc.make_inputs(case)
        rc_, co, rec, modes, taps = run_qp(ref, "ref_raht_inter_qp", p, True, morton, attrs, None, mref, aref, depth, rdo, fest, skip, q)
        assert rc_ == 0
        rc_, _, dec, _, _ = run_qp(ref, "ref_raht_inter_qp", p, False, morton, attrs, co, mref, aref, depth, rdo, fest, skip, q, modes, taps)
        assert rc_ == 0 and np.array_equal(dec, rec)
        out[name + "/in_sha"] = np.array(sha(morton, attrs, mref, aref) + ("" if q is None else sha(q)))
        out[name + "/coeffs"] = co
        out[name + "/rec"] = rec
        out[name + "/modes"] = modes
        out[name + "/taps"] = taps
        print(f"{name:22s} n={len(morton):5d} c={attrs.shape[1]} nonzero={np.count_nonzero(co):5d} modes={modes.tolist()} taps={taps.tolist()}")
    path = os.path.join(HERE, "raht_inter_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

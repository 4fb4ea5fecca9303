#!/usr/bin/env python3
"""Generate tests/golden/raht_golden.npz from the COMPILED REFERENCE
(oracle/_ref/libtmc3_ref.so, built by oracle/Makefile from the sources
under /root/reference).  Run in the build container:

    make -C oracle && python tests/golden/make_golden.py

For every case of tests/raht_cases.py the reference's forward outputs
(coefficients + reconstruction) and its inverse output are recorded: full
int32 arrays up to FULL_ARRAY_MAX_N points, SHA-256 digests above.  Inputs
are not stored; they are regenerated from the seeds."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402,F401  (registers the package alias)
import oracle_loader as ol  # noqa: E402
import raht_cases as rc  # noqa: E402


def main():
    ref = ol.ref()
    out = {}
    for case in rc.CASES:
        p, morton, attrs, qp = rc.make_inputs(case)
        n, c = attrs.shape
        coeffs, rec = ref.raht_forward(p, morton, attrs, qp)
        inv = ref.raht_inverse(p, morton, coeffs, c, qp)
        name = case["name"]
        out[name + "/n"] = np.array([n, c], dtype=np.int64)
        out[name + "/in_sha"] = np.array(rc.digest(morton) + rc.digest(attrs))
        out[name + "/sha"] = np.array(rc.digest(coeffs) + rc.digest(rec) + rc.digest(inv))
        if n <= rc.FULL_ARRAY_MAX_N:
            out[name + "/coeffs"] = coeffs
            out[name + "/rec"] = rec
            out[name + "/inv"] = inv
        print(f"{name:28s} n={n:7d} c={c} nonzero={np.count_nonzero(coeffs)}"
              f" enc_rec==dec_rec={np.array_equal(rec, inv)}")
    path = os.path.join(HERE, "raht_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

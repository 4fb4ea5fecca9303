#!/usr/bin/env python3
"""Generate tests/golden/recolour_golden.npz from the COMPILED REFERENCE (oracle/_ref/libtmc3_ref.so: pcc::recolour,
pointset_processing.cpp:926-957) for the cases of tests/recolour_cases.py.  Inputs are regenerated from the seeds (their
SHA-256 is stored).  Run in the build container:   make -C oracle && python tests/golden/make_recolour_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402,F401
import oracle_loader as ol  # noqa: E402
import recolour_cases as rc  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ref = ol.ref()
    out = {}
    for case in rc.CASES:
        p, xyz, a, tgt, scale = rc.make_inputs(case)
        got = ref.recolour(p, xyz, a, tgt, scale=scale)
        out[case[0] + "/in_sha"] = np.array(sha(xyz, a, tgt))
        out[case[0] + "/attrs"] = got.astype(np.int32)
        print(f"{case[0]:20s} source {len(xyz):5d} target {len(tgt):5d} c={a.shape[1]}")
    path = os.path.join(HERE, "recolour_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

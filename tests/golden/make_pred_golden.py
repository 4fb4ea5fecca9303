#!/usr/bin/env python3
"""Generate tests/golden/pred_golden.npz from the COMPILED REFERENCE at operator level (oracle/_ref: AttributeEncoder::
encode + AttributeDecoder::decode of the predicting transform, the symbols read back from the payload by the reference's
own entropy decoder) for a subset of the cases of tests/test_oracle_pred.py: the symbol stream (`values`), the
reconstruction, the inter-component prediction coefficients.  Inputs are regenerated from the seeds (their SHA-256 is
stored).  Run in the build container:   make -C oracle && python tests/golden/make_pred_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402,F401
import test_oracle_pred as tp  # noqa: E402

NAMES = ["dense_qp10", "dense_qp46_noicp", "dense_direct1", "dense_direct2", "dense_avg_disabled", "dense_nodirect_qnw",
         "dense_skip_intra", "dense_thr0", "lidar_refl_ctc", "lidar_refl_lods", "lidar_refl_direct2", "dense_scalable",
         "random_sparse", "tiny"]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    out = {}
    for name in NAMES:
        xyz, attrs, lod, pp, values, rec, icp = tp.run_reference(name)
        out[name + "/in_sha"] = np.array(sha(xyz, attrs))
        out[name + "/values"] = np.asarray(values, dtype=np.int32)
        out[name + "/rec"] = np.asarray(rec, dtype=np.int32)
        out[name + "/icp"] = np.asarray(icp, dtype=np.int32)
        print(f"{name:22s} n={len(xyz):6d} c={attrs.shape[1]} nonzero symbols={np.count_nonzero(values)}")
    path = os.path.join(HERE, "pred_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

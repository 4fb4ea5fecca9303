"""Worker of tests/test_gpu_raht_inter.py: loads oracle/_ref/libtmc3_shim.so (the reference's objects with the link
seams replaced, the HIP library inside) -- in a process of its own, never next to libtmc3_ref.so -- runs the
inter-frame RAHT cases through it and leaves the results in an .npz for the test to compare."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.load_package()


def counters(lib):
    out = (C.c_longlong * 2)()
    lib.gpcc_shim_raht_counters(out)
    return np.array([int(out[0]), int(out[1])])


def main():
    what, out = sys.argv[1], sys.argv[2]
    import oracle_loader as ol
    ol.ref = None  # (this process must not load the unmodified library)
    import test_gpu_raht_inter as t
    from test_oracle_raht_inter import _operator_roundtrip, run
    from mpeg_pcc_tmc13_amd import raht_params
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtmc3_shim.so"))
    res = {}
    if what == "function":
        morton, a_sorted, mref, aref, cases = t.seam1_cases()
        for i, (p, rdo, fest, _) in enumerate(cases):
            c0 = counters(lib)
            rc, co, rec, modes, taps = run(lib, "ref_raht_inter", p, True, morton, a_sorted, None, mref, aref, 15, rdo, fest, 3)
            assert rc == 0
            rc, _, dec, _, _ = run(lib, "ref_raht_inter", p, False, morton, a_sorted, co, mref, aref, 15, rdo, fest, 3, modes, taps)
            assert rc == 0
            res.update({f"co{i}": co, f"rec{i}": rec, f"modes{i}": modes, f"taps{i}": taps, f"dec{i}": dec,
                        f"calls{i}": counters(lib) - c0})
    else:
        xyz, attrs, xr, ar = t.operator_case()
        import lod_helpers as lh
        for i, (kw, rdo, fest, region) in enumerate(t.OPERATOR_CASES):
            c0 = counters(lib)
            lh.ref_set_qp_region(region, lib=lib)
            try:
                pay, enc, dec, modes, taps = _operator_roundtrip(raht_params(**kw), 34, xyz, attrs, xr, ar, 15, rdo, fest, 3, lib=lib)
            finally:
                lh.ref_set_qp_region(None, lib=lib)
            res.update({f"payload{i}": np.frombuffer(pay, np.uint8), f"enc{i}": enc, f"dec{i}": dec, f"modes{i}": modes,
                        f"taps{i}": taps, f"calls{i}": counters(lib) - c0})
    np.savez(out, **res)


if __name__ == "__main__":
    main()

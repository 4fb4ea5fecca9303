"""The two arithmetic back ends of the RAHT dependency kernels (csrc/raht_arith.hpp) against each
other on the CPU: ArithF64 -- doubles that hold integers, fma + trunc -- must give ArithI64's bits
(the reference's FixedPoint / Quantizer arithmetic, tmc3/FixedPoint.h:78-123, quantization.h:79-102)
for every primitive wherever its range conditions hold.  tests/emu/arith_check.cpp draws the cases."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_f64_primitives_give_the_int64_bits(tmp_path):
    exe = str(tmp_path / "arith_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-attributes", "-Wno-unknown-pragmas",
                    "-I", os.path.join(ROOT, "tests", "emu"), "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "mpeg-pcc-tmc13_amd", "csrc"),
                    os.path.join(ROOT, "tests", "emu", "arith_check.cpp"), "-o", exe], check=True)
    for seed in (1, 2, 3):
        r = subprocess.run([exe, "1000000", str(seed)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout

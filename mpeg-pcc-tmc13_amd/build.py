"""Build the gfx950 shared library IN-TREE with hipcc (cross-compiles without a
GPU).  The .so is git-ignored but travels to the GPU box with the snapshot."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libgpcc_attr_mi355.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".hpp"))) + [os.path.join(ROOT, "include", "gpcc_attr_mi355.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           *os.environ.get("GPCC_EXTRA_FLAGS", "").split(),
           os.path.join(CSRC, "gpcc_attr_mi355.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

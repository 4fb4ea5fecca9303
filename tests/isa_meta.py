"""Kernel descriptors of the BUILT library (no compile: the gfx950 code object is unbundled from the .so's
.hip_fatbin section and its amdhsa metadata notes are read): {kernel name: {vgpr, sgpr, scratch, lds}}."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "mpeg-pcc-tmc13_amd", "libgpcc_attr_mi355.so")


def available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-readelf", "clang-offload-bundler")) and os.path.exists(LIB)


def kernel_meta(lib=LIB):
    sec = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", lib], capture_output=True, text=True,
                         check=True).stdout
    off = size = None
    for ln in sec.splitlines():
        if ".hip_fatbin" in ln:
            f = ln.split()
            i = f.index(".hip_fatbin")
            off, size = int(f[i + 3], 16), int(f[i + 4], 16)
    assert off is not None, "no .hip_fatbin section"
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        with open(lib, "rb") as f:
            f.seek(off)
            open(fat, "wb").write(f.read(size))
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                               check=True).stdout
    out, cur = {}, {}
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k in cur:  # the next kernel's record begins
            if "name" in cur and "vgpr_count" in cur:
                out[cur["name"]] = cur
            cur = {}
        cur[k] = v if k == "name" else int(v)
    if "name" in cur and "vgpr_count" in cur:
        out[cur["name"]] = cur
    return {n: {"vgpr": d["vgpr_count"], "sgpr": d.get("sgpr_count", 0), "scratch": d.get("private_segment_fixed_size", 0),
                "lds": d.get("group_segment_fixed_size", 0)} for n, d in out.items()}


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    out = subprocess.run([filt] + list(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))

#!/usr/bin/env python3
"""Per-kernel summary of the gfx950 code object: registers, scratch, occupancy
and the instruction classes that matter for the latency-bound kernels of this
library -- vector-memory loads, `s_waitcnt vmcnt` (on gfx9 it also waits for
stores in flight), loads that are waited for at once (a dependent round trip
each), `ds_bpermute` (LDS-crossbar shuffles) vs DPP moves, flat accesses.

    python tools/isa_audit.py > profiles/rNN_isa_audit.txt

Compiles mpeg-pcc-tmc13_amd/csrc/gpcc_attr_mi355.hip with `-S`; no GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mpeg-pcc-tmc13_amd", "csrc", "gpcc_attr_mi355.hip")


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return {n: n for n in names}
    out = subprocess.run([filt] + names, capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC),
                        "-S", "--cuda-device-only", "-o", asm, SRC] + sys.argv[1:], check=True)
        text = open(asm).read().splitlines()
    kernels, cur = {}, None
    for ln in text:
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = {"body": [], "meta": {}}
            continue
        if cur is None:
            continue
        m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", ln)
        if m:
            kernels[cur]["meta"][m.group(1)] = int(m.group(2))
            if m.group(1) == "Occupancy":
                cur = None
            continue
        kernels[cur]["body"].append(ln.strip())
    names = demangle(list(kernels))
    rows = []
    for k, v in kernels.items():
        if "Occupancy" not in v["meta"]:
            continue
        ins = [b for b in v["body"] if b and not b.startswith((";", ".")) and not b.endswith(":")]
        is_load = lambda s: re.match(r"(global_load|buffer_load|flat_load|scratch_load)", s) is not None
        loads = sum(is_load(s) for s in ins)
        stores = sum(re.match(r"(global_store|buffer_store|flat_store)", s) is not None for s in ins)
        waits = sum(s.startswith("s_waitcnt") and "vmcnt" in s for s in ins)
        at_once = 0
        for i, s in enumerate(ins):
            if is_load(s) and any("vmcnt(0)" in t for t in ins[i + 1:i + 4]):
                at_once += 1
        rows.append((names[k].replace("void ", "").replace("gpcc::", "")[:64], v["meta"].get("NumVgprs", 0),
                     v["meta"].get("ScratchSize", 0), v["meta"].get("Occupancy", 0),
                     v["meta"].get("LDSByteSize", 0), len(ins), loads, at_once, waits, stores,
                     sum("ds_bpermute" in s for s in ins), sum("_dpp" in s for s in ins),
                     sum(s.startswith("flat_") for s in ins), sum(s.startswith("s_load") for s in ins)))
    hdr = ("kernel", "vgpr", "scratch", "occ", "lds", "insts", "vloads", "waited_at_once", "vmcnt_waits",
           "vstores", "bpermute", "dpp", "flat", "s_load")
    print("# static counts per kernel (whole body, loops counted once); occ = waves/SIMD")
    print("%-64s %5s %7s %3s %6s %6s %6s %14s %11s %7s %8s %4s %4s %6s" % hdr)
    for r in sorted(rows, key=lambda r: -r[5]):
        print("%-64s %5d %7d %3d %6d %6d %6d %14d %11d %7d %8d %4d %4d %6d" % r)
    # an exactly-fitting 56-register wavefront is what the LDS form of finish_kernel failed in on the
    # MI355X (profiles/r04_finish_lds_root_cause.txt): a kernel that lands there WITH tables in LDS pads
    # its allocation (asm volatile("" ::: "v63"))
    a56 = [r for r in rows if (r[1] + 7) // 8 * 8 == 56]
    print("# kernels with a 56-register allocation: " + (", ".join("%s (lds %d)" % (r[0], r[4]) for r in a56) or "none"))


if __name__ == "__main__":
    main()

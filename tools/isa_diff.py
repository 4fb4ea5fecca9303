#!/usr/bin/env python3
"""Which device kernels changed between two commits?  Compiles gpcc_attr_mi355.hip of both trees to
gfx950 assembly (device only) and compares every kernel's instruction stream (labels and comments
normalised).  Used at the end of round 3 to show that the work done after the last run on an MI355X
left every kernel that had run there untouched:      python tools/isa_diff.py <old commit> [new commit]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(commit, td):
    src = os.path.join(td, commit)
    os.makedirs(src)
    tar = subprocess.run(["git", "-C", ROOT, "archive", commit, "mpeg-pcc-tmc13_amd/csrc", "include"], capture_output=True, check=True).stdout
    subprocess.run(["tar", "-x", "-C", src], input=tar, check=True)
    out = os.path.join(td, commit + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(src, "include"),
                    "-I" + os.path.join(src, "mpeg-pcc-tmc13_amd/csrc"), "-S", "--cuda-device-only", "-o", out,
                    os.path.join(src, "mpeg-pcc-tmc13_amd/csrc/gpcc_attr_mi355.hip")], check=True)
    kernels, cur = {}, None
    for ln in open(out):
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None:
            continue
        if ln.startswith("; Occupancy"):
            cur = None
            continue
        s = ln.strip()
        if s and not s.startswith((";", ".")):
            kernels[cur].append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", s).strip()))
    return kernels


def main():
    old = sys.argv[1]
    new = sys.argv[2] if len(sys.argv) > 2 else "HEAD"
    with tempfile.TemporaryDirectory() as td:
        a, b = assembly(old, td), assembly(new, td)
    both = [k for k in a if k in b]
    changed = [k for k in both if a[k] != b[k]]
    names = subprocess.run(["c++filt"] + changed + [k for k in a if k not in b] + [k for k in b if k not in a],
                           capture_output=True, text=True).stdout.splitlines()
    print(f"{old} -> {new}: {len(both)} kernels in both, {len(both) - len(changed)} with identical instructions, {len(changed)} changed")
    for k, nm in zip(changed, names):
        d = [(x, y) for x, y in zip(a[k], b[k]) if x != y]
        print(f"  changed  {nm[:100]}  {len(a[k])} -> {len(b[k])} instructions, {len(d)} differing lines; first: {d[0] if d else 'length only'}")
    rest = names[len(changed):]
    gone = [k for k in a if k not in b]
    for nm in rest[:len(gone)]:
        print(f"  only in {old}: {nm[:110]}")
    for nm in rest[len(gone):]:
        print(f"  new in {new}: {nm[:110]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc results (the rocpd sqlite output).

    python tools/pmc_summary.py <results.db> [<results.db> ...]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE
counts 1/2 of a wide coalesced stream (MI355X_MICROARCH.md)."""
import collections
import sqlite3
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        seen = set()
        for name, disp, cname, val, d in db.execute(
                "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
            k = name.split("(")[0].replace("void gpcc::", "").replace("gpcc::", "")
            acc[k][cname] += val
            launches[k].add((path, disp))
            if (path, disp) not in seen:
                seen.add((path, disp))
                dur[(k, path)] += d
    order = sorted(acc, key=lambda k: -max([v for (kk, _), v in dur.items() if kk == k] or [0]))
    for k in order:
        n = len({d for _, d in launches[k]})
        tot = max(v for (kk, _), v in dur.items() if kk == k)
        print(f"{k}  launches {n}  total {tot / 1e3:.1f} us (under the profiler)")
        c = acc[k]
        for name in sorted(c):
            print(f"    {name:28s} {c[name]:.6g}")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            print("    " + "  ".join(f"{nm[3:]}/WAVE_CYCLES {c[nm] / wc:.3f}" for nm in
                                     ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS")
                                     if nm in c))
        if "SQ_WAVES" in c:
            print("    per wave: " + "  ".join(f"{nm[8:]} {c[nm] / c['SQ_WAVES']:.0f}" for nm in sorted(c)
                                               if nm.startswith("SQ_INSTS_")))


if __name__ == "__main__":
    main()

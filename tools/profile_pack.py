#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scratch-free profiling runs into the summaries
kept under profiles/:

    python tools/profile_pack.py stats <kernel_stats.csv> "<header line>" ...   -> fixed-width table on stdout
    python tools/profile_pack.py pmc <pass1.txt> <pass2.txt> ...                 -> per-kernel merge of
        tools/pmc_summary.py outputs of SEPARATE --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*), ratios recomputed
"""
import collections
import csv
import re
import sys


def stats(path, headers):
    for h in headers:
        print("# " + h)
    print(f"{'kernel':78s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
    for r in csv.DictReader(open(path)):
        print(f"{r['Name'][:78]:78s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / 1e3:12.1f} "
              f"{float(r['AverageNs']) / 1e3:10.2f} {int(r['MinNs']) / 1e3:9.2f} {int(r['MaxNs']) / 1e3:10.2f} "
              f"{float(r['Percentage']):6.2f}")


def pmc(paths):
    acc = collections.OrderedDict()
    for p in paths:
        cur = None
        for line in open(p):
            m = re.match(r"^(\S.*?)  launches (\d+)  total ([\d.]+) us", line)
            if m:
                cur = acc.setdefault(m.group(1), dict(launches=int(m.group(2)), total=float(m.group(3)), c={}))
                continue
            m = re.match(r"^\s+([A-Z_0-9]+)\s+([-+.\deE]+)\s*$", line)
            if m and cur is not None:
                cur["c"][m.group(1)] = float(m.group(2))
    for k, v in acc.items():
        if "gpcc" not in k and "raht" not in k and "tile" not in k and "tree" not in k and "rdoq" not in k \
                and "lod" not in k and "finish" not in k and "schedule" not in k and "ascend" not in k:
            continue
        print(f"{k}  launches {v['launches']}  total {v['total']:.1f} us (under the profiler)")
        c = v["c"]
        for name in sorted(c):
            print(f"    {name:28s} {c[name]:.6g}")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            print("    " + "  ".join(f"{nm[3:]}/WAVE_CYCLES {c[nm] / wc:.3f}" for nm in
                                     ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if nm in c))
        if "SQ_WAVES" in c:
            print("    per wave: " + "  ".join(f"{nm[8:]} {c[nm] / c['SQ_WAVES']:.0f}" for nm in sorted(c)
                                               if nm.startswith("SQ_INSTS_")))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            print(f"    per launch: FETCH {c['FETCH_SIZE'] * 1024 / v['launches']:.0f} B  "
                  f"WRITE {c['WRITE_SIZE'] * 1024 / v['launches']:.0f} B (raw counter bytes)")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3:])
    else:
        pmc(sys.argv[2:])

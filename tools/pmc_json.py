#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of rocprofv3 (rocpd .db files) -> {kernel: {fetch_bytes_per_launch,
write_bytes_per_launch, launches}} for one workload (raw counter bytes: rocprofv3 reports KiB).

    python tools/pmc_json.py <workload> <results.db> [...]  > traffic_<workload>.json"""
import collections
import json
import sqlite3
import sys


def timer_name(k):
    """the name the library's own per-kernel timers (gpcc_ctx_kernel_times, bench.py's roofline objects) use"""
    import re
    m = re.match(r"raht_level_sub_kernel<\d+, (\d)", k)
    if m:
        return {"1": "level_sub_synth", "2": "level_sub_fused", "3": "level_sub_lossy"}[m.group(1)]
    m = re.match(r"raht_sub_sweep_kernel<\d+, (\d)", k)
    if m:
        return {"1": "sub_sweep_synth", "3": "sub_sweep_lossy"}[m.group(1)]
    m = re.match(r"cx_level_kernel<\d+, (true|false)", k)
    if m:
        return "cx_level_enc" if m.group(1) == "true" else "cx_level_dec"
    m = re.match(r"cx_top_kernel<\d+, (true|false)", k)
    if m:
        return "cx_top_enc" if m.group(1) == "true" else "cx_top_dec"
    k = k.split("<")[0]
    for a, b in (("raht_level_prepass_kernel", "level_prepass"), ("_kernel", "")):
        k = k.replace(a, b)
    return k


def main():
    workload = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        for name, disp, cname, val in db.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            k = timer_name(name.split("(")[0].replace("void gpcc::", "").replace("gpcc::", ""))
            acc[k][cname] += val
            launches[k][cname].add((path, disp))
    out = {}
    for k, c in acc.items():
        nf = len(launches[k].get("FETCH_SIZE", ())) or 1
        nw = len(launches[k].get("WRITE_SIZE", ())) or 1
        out[k] = {"fetch_bytes_per_launch": round(c.get("FETCH_SIZE", 0.0) * 1024 / nf),
                  "write_bytes_per_launch": round(c.get("WRITE_SIZE", 0.0) * 1024 / nw),
                  "launches": max(nf, nw)}
    print(json.dumps({workload: out}, indent=1))


if __name__ == "__main__":
    main()

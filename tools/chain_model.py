#!/usr/bin/env python3
"""Offline model of the dependency chains of RAHT with sub-node prediction
(DESIGN.md sections 5 and 7): per octree level, the parents in Morton order,
the 12 causal neighbour directions whose children a block may use
(tmc3/RAHT.cpp:314-326: the three negative faces, the three negative edges and
six mixed edges), and the longest chain -- in hops, and in hops that cross a
boundary when G (wavefront claim) or WG (workgroup) consecutive blocks are
grouped.  An upper bound: every existing preceding neighbour counts, whether
or not the child it would contribute is occupied.

    python tools/chain_model.py lidar|dense [G=8] [WG=32]        (CPU only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from mpeg_pcc_tmc13_amd import synth  # noqa: E402

OFFS = [(-1, -1, 0), (-1, 0, -1), (-1, 0, 0), (0, -1, -1), (0, -1, 0), (0, 0, -1),
        (-1, 0, 1), (0, -1, 1), (-1, 1, 0), (0, 1, -1), (1, -1, 0), (1, 0, -1)]


def part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    for sh, mask in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f),
                     (4, 0x10c30c30c30c30c3), (2, 0x1249249249249249)):
        v = (v | (v << np.uint64(sh))) & np.uint64(mask)
    return v


def morton(x, y, z):
    return (part1by2(x) << np.uint64(2)) | (part1by2(y) << np.uint64(1)) | part1by2(z)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lidar"
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    WG = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    xyz = (synth.lidar_cloud(1_000_000, seed=1) if kind == "lidar" else synth.dense_cloud(1_000_000, seed=1, bits=10))[0]
    tot = np.zeros(4, np.int64)
    for lvl in range(1, int(xyz.max()).bit_length() + 1):  # parents of size 2^lvl
        p = np.unique(xyz >> lvl, axis=0)
        key = morton(p[:, 0], p[:, 1], p[:, 2])
        o = np.argsort(key)
        key, p = key[o], p[o]
        m = len(key)
        if m < 2:
            continue
        dep = np.full((m, 12), -1, np.int64)
        for i, d in enumerate(OFFS):
            q = p + np.array(d)
            ok = (q >= 0).all(1)
            q = np.maximum(q, 0)
            qk = morton(q[:, 0], q[:, 1], q[:, 2])
            idx = np.minimum(np.searchsorted(key, qk), m - 1)
            hit = ok & (key[idx] == qk) & (idx < np.arange(m))
            dep[hit, i] = idx[hit]
        da, dg, dw = [0] * m, [0] * m, [0] * m
        for j, row in enumerate(dep.tolist()):
            a = g = w = 0
            for q in row:
                if q >= 0:
                    a = max(a, da[q] + 1)
                    g = max(g, dg[q] + (q // G != j // G))
                    w = max(w, dw[q] + (q // WG != j // WG))
            da[j], dg[j], dw[j] = a, g, w
        row = (m, max(da), max(dg), max(dw))
        tot += row
        print(f"{kind} level {lvl:2d}: parents {row[0]:7d}  depth {row[1]:5d}  crossing {G}-block groups {row[2]:5d}  "
              f"crossing {WG}-block groups {row[3]:5d}")
    print(f"{kind} TOTAL: parents {tot[0]}  depth {tot[1]}  crossing {G}: {tot[2]}  crossing {WG}: {tot[3]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Offline model: what walking the sub-node dependency DAG ACROSS levels would
buy.  A block (parent with >= 2 children) costs one hop
and needs (a) the values of its own node and of the parent-level neighbours it
predicts from -- each produced by the block one level up that owns that node --
and (b) the blocks of the 12 causal same-level neighbours.  A parent with one
child is a copy (no hop).  Prints the longest path of that DAG next to the sum
of the per-level depths (levels launched one after the other).

    python tools/chain_model_xlevel.py lidar|dense        (CPU only, a few minutes)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from chain_model import OFFS, morton  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from mpeg_pcc_tmc13_amd import synth  # noqa: E402

ALL26 = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)
         if (dx, dy, dz) != (0, 0, 0) and abs(dx) + abs(dy) + abs(dz) <= 2]  # 18 face / edge neighbours


def lookup(key, p, d):
    q = p + np.array(d)
    ok = (q >= 0).all(1)
    q = np.maximum(q, 0)
    qk = morton(q[:, 0], q[:, 1], q[:, 2])
    idx = np.minimum(np.searchsorted(key, qk), len(key) - 1)
    hit = ok & (key[idx] == qk)
    return np.where(hit, idx, -1)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lidar"
    xyz = (synth.lidar_cloud(1_000_000, seed=1) if kind == "lidar" else synth.dense_cloud(1_000_000, seed=1, bits=10))[0]
    top = int(xyz.max()).bit_length()
    levels = {}
    for lvl in range(0, top + 1):  # nodes of size 2^lvl (lvl 0 = the points' voxels)
        p = np.unique(xyz >> lvl, axis=0)
        key = morton(p[:, 0], p[:, 1], p[:, 2])
        o = np.argsort(key)
        levels[lvl] = (key[o], p[o])
    # avail[lvl][n]: hop count at which node n of size 2^lvl has its value; root: 0
    avail = {top: np.zeros(len(levels[top][0]), np.int64)}
    seq_total = 0
    for lvl in range(top, 0, -1):  # blocks: parents of size 2^lvl, children of size 2^(lvl-1)
        key, p = levels[lvl]
        ckey, cp = levels[lvl - 1]
        m = len(key)
        # children count per parent
        par_of_child = np.searchsorted(key, morton(cp[:, 0] >> 1, cp[:, 1] >> 1, cp[:, 2] >> 1))
        nchild = np.bincount(par_of_child, minlength=m)
        multi = nchild >= 2
        same = np.stack([np.where((idx := lookup(key, p, d)) < np.arange(m), idx, -1) for d in OFFS], 1)
        nb = np.stack([lookup(key, p, d) for d in ALL26], 1)
        av = avail[lvl]
        base = av.copy()  # own node
        nbav = np.where(nb >= 0, av[np.maximum(nb, 0)], 0).max(1)
        base = np.where(multi, np.maximum(base, nbav), base)
        T = [0] * m      # cross-level finish time
        D = [0] * m      # per-level depth (levels one after the other)
        mul = multi.tolist()
        bl = base.tolist()
        for j, row in enumerate(same.tolist()):
            if not mul[j]:
                T[j] = bl[j]
                continue
            t = bl[j]
            dd = 0
            for q in row:
                if q >= 0:
                    if T[q] > t:
                        t = T[q]
                    if mul[q] and D[q] > dd:
                        dd = D[q]
            T[j] = t + 1
            D[j] = dd + 1
        T = np.array(T, np.int64)
        avail[lvl - 1] = T[par_of_child]
        seq_total += max(D)
        print(f"{kind} blocks of size 2^{lvl:2d}: {m:7d} ({int(multi.sum()):7d} with >= 2 children)  depth alone {max(D):5d}  "
              f"finish across levels {int(T.max()):6d}")
    print(f"{kind}: levels one after the other {seq_total} hops; across levels {int(avail[0].max())} hops")


if __name__ == "__main__":
    main()

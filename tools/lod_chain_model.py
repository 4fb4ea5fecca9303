#!/usr/bin/env python3
"""Offline critical path of the distance sub-sampler (tmc3/PCCTMC3Common.h:1984-2085) per level of detail.

At most one point of a cell is retained, and whether a cell retains one depends on the retained points of the
19 causal neighbour cells (same 128^3 atlas block): a DAG over the cells of a level.  This script replays the
reference's greedy order on the bench's lifting workload (1 M-point S-dense slice, dist2 from lod_params()) and
reports, per level: cells, retained points, and the longest dependency path
  depth(c) = 1 + max depth(n) over the causal neighbour cells n that hold input points
(the kernel drops neighbours that cannot reach any candidate of the cell -- exact pruning, lod_kernels.hpp --
so its own chains are shorter: `depth_reach` applies the same test).  With the per-level kernel times of
tools/lod_level_times.py this gives the time per hop, i.e. how far the kernel is from its dependency floor.

    python tools/lod_chain_model.py [points] > profiles/r06_lod_chain_model.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
from mpeg_pcc_tmc13_amd import lod_params, synth  # noqa: E402

K_ATLAS = 21
OFFS = [7, 3, 5, 6, 12, 10, 17, 20, 34, 33, 4, 2, 1, 24, 40, 48, 32, 16, 8, 0]  # (:2013-2034), own cell first


def spread3(v):
    r = 0
    for b in range(21):
        r |= ((v >> b) & 1) << (3 * b)
    return r


def compact3(m):
    r = 0
    for b in range(21):
        r |= ((m >> (3 * b)) & 1) << b
    return r


def cell_xyz(cell):
    return compact3(cell >> 2), compact3(cell >> 1), compact3(cell)


def neighbours(cell):
    x, y, z = cell_xyz(cell)
    out = []
    for off in OFFS[1:]:
        dx = ((off >> 2) & 1) + 2 * ((off >> 5) & 1) - 1
        dy = ((off >> 1) & 1) + 2 * ((off >> 4) & 1) - 1
        dz = (off & 1) + 2 * ((off >> 3) & 1) - 1
        nx, ny, nz = x + dx, y + dy, z + dz
        if nx < 0 or ny < 0 or nz < 0:
            continue
        out.append(((spread3(nx) << 2) | (spread3(ny) << 1) | spread3(nz), (dx, dy, dz)))
    return out


def level(codes, xyz, inp, shift0):
    """one call of subsampleByDistance: -> retained indices, stats"""
    if len(inp) == 1:
        return [], dict(cells=1, retained=0, depth=1, depth_reach=1)
    radius2 = 3 << (2 * shift0)
    sh = shift0 + 1
    shift3 = 3 * sh
    edge = 1 << sh
    cells = [int(c) >> shift3 for c in codes[inp]]
    # the cells of the level in order: first point, point count
    starts = [0] + [i for i in range(1, len(cells)) if cells[i] != cells[i - 1]] + [len(cells)]
    depth, depth_r, kept = {}, {}, {}
    retained = []
    for a, b in zip(starts[:-1], starts[1:]):
        c = cells[a]
        atlas = c >> K_ATLAS
        pts = [xyz[inp[t]] for t in range(a, b)]
        nbs = [(n, d) for n, d in neighbours(c) if (n >> K_ATLAS) == atlas and n < c and n in depth]
        # the reference's greedy: the first point not within the radius of a retained point of a neighbour
        keep = -1
        for u, p in enumerate(pts):
            hit = False
            for n, _ in nbs:
                q = kept.get(n)
                if q is not None and int((q[0] - p[0])) ** 2 + int((q[1] - p[1])) ** 2 + int((q[2] - p[2])) ** 2 <= radius2:
                    hit = True
                    break
            if not hit:
                keep = u
                break
        if keep >= 0:
            kept[c] = pts[keep]
            retained.append(inp[a + keep])
        depth[c] = 1 + max([depth[n] for n, _ in nbs], default=0)
        # reach pruning: a neighbour cell none of whose voxels lies within the radius of any candidate cannot matter
        x0, y0, z0 = [v * edge for v in cell_xyz(c)]
        dr = 0
        for n, (dx, dy, dz) in nbs:
            lx, ly, lz = x0 + dx * edge, y0 + dy * edge, z0 + dz * edge
            for p in pts[:8]:
                gx = max(lx - p[0], p[0] - (lx + edge - 1), 0)
                gy = max(ly - p[1], p[1] - (ly + edge - 1), 0)
                gz = max(lz - p[2], p[2] - (lz + edge - 1), 0)
                if gx * gx + gy * gy + gz * gz <= radius2:
                    dr = max(dr, depth_r[n])
                    break
        depth_r[c] = 1 + dr
    return retained, dict(cells=len(starts) - 1, retained=len(retained), depth=max(depth.values()),
                          depth_reach=max(depth_r.values()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    xyz, col = synth.dense_cloud(n, seed=201, bits=10)
    codes, _, order = synth.sort_by_morton(xyz, col)
    xyz = xyz[order].astype(np.int64)
    lp = lod_params()
    inp = np.arange(len(codes))
    print(f"# distance sub-sampler, {len(codes)}-point S-dense slice, dist2 shift {lp.dist2}+{lp.attr_dist2_delta}, "
          f"{lp.num_detail_levels_minus1 + 1} levels of detail")
    print("# lod  input points      cells   retained   depth (all causal cells)   depth (cells that can reach a candidate)")
    total = total_r = 0
    for lod in range(lp.num_detail_levels_minus1):
        if len(inp) <= 1:
            break
        ret, st = level(codes, xyz, inp, lp.dist2 + lp.attr_dist2_delta + lod)
        print(f"{lod:5d} {len(inp):13d} {st['cells']:10d} {st['retained']:10d} {st['depth']:12d} {st['depth_reach']:28d}")
        total += st["depth"]
        total_r += st["depth_reach"]
        inp = np.array(ret, dtype=np.int64)
    print(f"# sum of the levels' depths: {total} (all causal cells), {total_r} (reach-pruned)")


if __name__ == "__main__":
    main()

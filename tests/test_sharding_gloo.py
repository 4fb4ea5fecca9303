"""CPU-only, gloo, world sizes 2 and 8 (the node's size): the N>1 path of
bench.py -- units sharded across ranks, transformed locally (here by the CPU
oracle standing in for the GPU), coefficients gathered on rank 0 -- equals
the single-process result unit by unit.  Ragged units, ranks with several
units, ranks with NONE (3 units on 8 ranks: five ranks send an empty
buffer), and configs[4]'s shape: ten ragged slices on eight ranks (2,2,1,1,1,1,1,1)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sizes, result_path):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    import oracle_loader as ol
    from mpeg_pcc_tmc13_amd import raht_params, sharding, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = raht_params(qp=30, subnode=False)
    mine = sharding.shard_units(sizes, world)[rank]
    parts = []
    for u in mine:
        xyz, col = synth.random_cloud(sizes[u], seed=100 + u, bits=5)
        m, a, _ = synth.sort_by_morton(xyz, col)
        co, _ = ol.oracle().raht_forward(p, m, a)
        parts.append(co)
    local = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.int32))
    got = sharding.gather_coefficients(local, dst=0)
    if rank == 0:
        assign = sharding.shard_units(sizes, world)
        ok = True
        for r in range(world):
            off = 0
            for u in assign[r]:
                xyz, col = synth.random_cloud(sizes[u], seed=100 + u, bits=5)
                m, a, _ = synth.sort_by_morton(xyz, col)
                co, _ = ol.oracle().raht_forward(p, m, a)
                ok &= bool(np.array_equal(got[r][off:off + co.size].numpy(), co))
                off += co.size
            ok &= off == got[r].numel()
        open(result_path, "w").write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_units_balanced_and_deterministic():
    from mpeg_pcc_tmc13_amd import sharding
    sizes = [1100000, 900000, 1000000, 50, 700000, 1100000, 3, 800000]
    a = sharding.shard_units(sizes, 4)
    assert a == sharding.shard_units(sizes, 4)
    assert sorted(i for r in a for i in r) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in r) for r in a]
    assert max(loads) - min(loads) <= max(sizes)
    assert sharding.shard_units([5, 5], 4) == [[0], [1], [], []]


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process(tmp_path):
    sizes = [1500, 400, 2200, 1, 900]   # ragged, one rank gets more units
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, sizes, result), nprocs=2, join=True)
    assert open(result).read() == "ok"


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sizes", [
    [700, 1, 1300],                                         # 3 units on 8 ranks: five ranks hold nothing
    [900, 1100, 1000, 950, 1050, 1000, 980, 1020, 990, 1010],  # configs[4]: ten slices -> 2,2,1,1,1,1,1,1
], ids=["empty_ranks", "ten_slices"])
def test_eight_rank_gather_matches_single_process(tmp_path, sizes):
    from mpeg_pcc_tmc13_amd import sharding
    assign = sharding.shard_units(sizes, 8)
    assert sorted(u for r in assign for u in r) == list(range(len(sizes)))
    if len(sizes) == 3:
        assert sum(1 for r in assign if not r) == 5
    if len(sizes) == 10:
        assert sorted(len(r) for r in assign) == [1, 1, 1, 1, 1, 1, 2, 2]
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(8, port, sizes, result), nprocs=8, join=True)
    assert open(result).read() == "ok"

"""Multi-GPU layer: slices / frames are independent attribute-coding units
(reference tmc3/encoder.cpp:544-571 codes them one after another), so they
shard across ranks with NO data-path collective; the only exchange is one
gather of the quantised coefficient buffers to the rank that owns the
(serial, CPU) arithmetic coder.  One process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the node, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_units(sizes, world):
    """Deterministic size-balanced assignment (longest-processing-time
    first, ties by unit index).  Returns world lists of unit indices, each
    in ascending order."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(u) for u in out]


def gather_coefficients(coeffs, dst=0, group=None):
    """Gather one 1-D int32 coefficient tensor per rank on `dst` (lengths may
    differ).  Returns the list of per-rank tensors on dst, None elsewhere.
    7 peers -> 7 distinct xGMI links into dst: per-link bound, one message
    per peer."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([coeffs.numel()], dtype=torch.int64, device=coeffs.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    if coeffs.numel() == cap:
        send = coeffs.contiguous()
    else:
        send = torch.zeros(cap, dtype=coeffs.dtype, device=coeffs.device)
        send[:coeffs.numel()] = coeffs
    recv = [torch.empty(cap, dtype=coeffs.dtype, device=coeffs.device) for _ in range(world)] \
        if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[r][:sizes[r]] for r in range(world)]

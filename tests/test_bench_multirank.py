"""bench.py with two ranks: the driver launches it under torch.distributed.run,
one rank per GPU over RCCL.  On a one-GPU box the same control flow (rendezvous
on 127.0.0.1, per-rank frames, gather of the coefficient buffers on rank 0,
barrier, MAX over ranks, one JSON line from rank 0) is exercised with both ranks
sharing the device and the gather going through host memory
(GPCC_BENCH_BACKEND=gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(world, points, c3, c4):
    env = dict(os.environ, GPCC_BENCH_BACKEND="gloo")
    # (--verify-gather is the default for N > 1 since round 5)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--points", str(points), "--configs3-points", str(c3), "--configs4-points", str(c4),
           "--frames-per-gpu-batched", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_eight_ranks_share_the_gpu():
    """the node's size: eight ranks (here sharing one device, gather through host memory) -- rendezvous, per-rank
    frames, the gather, configs[3]'s frames and configs[4]'s ten slices sharded 2,2,1,1,1,1,1,1"""
    d = run_bench(8, 60000, 50000, 40000)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["distributed"]["world_size"] == 8
    assert d["config"]["roundtrip_decoder_equals_encoder_recon"] is True
    assert d["config"]["gathered_equals_single_rank"] is True
    assert d["configs3"]["roundtrip_decoder_equals_encoder_recon"] is True
    c4 = d["configs4"]
    assert c4["scaling"] == "strong" and sorted(c4["slices_per_rank"]) == [1, 1, 1, 1, 1, 1, 2, 2]
    assert c4["ceiling"].startswith("at most 5.00x")
    assert c4["roundtrip_decoder_equals_encoder_recon"] is True and c4["gathered_equals_single_rank"] is True
    assert c4["value"] > 0
    # the throughput regime of the weak-scaling line: several frames per GPU and step, per-rank device and gather times
    wb = d["weak_batched"]
    assert wb["frames_per_gpu"] == 3 and wb["scaling"] == "weak" and wb["value"] > 0
    assert wb["roundtrip_decoder_equals_encoder_recon"] is True
    assert len(wb["device_ms_per_rank"]) == 8 and len(wb["gather_ms_per_rank"]) == 8
    assert len(d["distributed"]["device_ms_per_rank"]) == 8 and min(d["distributed"]["device_ms_per_rank"]) > 0


@pytest.mark.gpu
def test_bench_two_ranks_one_json_line():
    d = run_bench(2, 200000, 150000, 100000)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["points_per_gpu_per_step"] == 200000
    assert d["configs4"]["slices_per_rank"] == [5, 5] and d["configs4"]["gathered_equals_single_rank"] is True
    assert d["config"]["roundtrip_decoder_equals_encoder_recon"] is True
    assert d["value"] > 0 and "roofline" not in d  # per-kernel figures are an N=1 report
    # the coefficient buffers rank 0 gathered are what one GPU computes for the same frames
    assert d["config"]["gathered_equals_single_rank"] is True
    # BASELINE configs[3] shape (dense colour frames, one per rank) in the same line
    assert d["configs3"]["roundtrip_decoder_equals_encoder_recon"] is True and d["configs3"]["value"] > 0
    assert d["weak_batched"]["frames_per_gpu"] == 3 and len(d["weak_batched"]["device_ms_per_rank"]) == 2
    assert len(d["distributed"]["gather_ms_per_rank"]) == 2

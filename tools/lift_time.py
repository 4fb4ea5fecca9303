#!/usr/bin/env python3
"""Lifting / LoD-build leg of bench.py alone (GPU box)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as g
g.load_package()
import torch
from mpeg_pcc_tmc13_amd import context
class A: points = 1_000_000; no_cpu_baseline = True
ctx = context(0)
print(json.dumps(bench.lifting_leg(ctx, A(), torch=torch, dev=torch.device("cuda:0"))))

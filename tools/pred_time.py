#!/usr/bin/env python3
"""Predicting-transform legs of bench.py alone (GPU box)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as g
g.load_package()
from mpeg_pcc_tmc13_amd import context
class A: points = 1_000_000; no_cpu_baseline = False
ctx = context(0)
print(json.dumps(bench.predicting_leg(ctx, A())))

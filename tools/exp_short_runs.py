#!/usr/bin/env python3
"""How much of a SHORT timed run is fixed cost?  bench.py's timed region (barrier + sync, K launches enqueued from C with events,
sync) for K = 5 .. 2000 steps: wall us per step and the implied fixed cost next to the 2000-step figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
n = 65536
eng = OvercookedEngine(get_base_layout_params("cramped_room", 200), n, 0)
acts = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
eng.rollout(acts, 200); torch.cuda.synchronize()
base = None
for K in (2000, 5, 20, 20, 20, 100, 500, 2000):
    eng.rollout(acts, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = eng.rollout_timed(acts, K)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    if K == 2000: base = wall / K
    print(f"K={K:5d}: wall {wall:9.1f} us = {wall / K:7.2f} us/step; events {ms * 1e3 / K:6.2f} us/step; fixed cost vs the long run {wall - K * (base or 0):7.1f} us")

#!/usr/bin/env python3
"""HIP_FORCE_DEV_KERNARG=0 / 1 (kernel arguments in host-coherent or in device memory): small policy launches and env steps.  Run once per setting."""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
from bench_fused_policy import random_net
print("HIP_FORCE_DEV_KERNARG =", os.environ.get("HIP_FORCE_DEV_KERNARG"))
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[random_net(5, 4, 6, ACTOR_HEAD, gen) for _ in range(2)])
def timed(fn, iters=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
for rows in (1000, 2000, 131072):
    obs = (torch.rand((rows, 5, 4, 20), device="cuda") < 0.1).to(torch.int8)
    plan = PolicyPlan(nets, rows, [(0, rows, 0, TILE_ACT)])
    act = torch.zeros(rows, dtype=torch.int32, device="cuda")
    print(f"policy {rows:7d} rows: {timed(lambda: plan.forward(obs, actions=act, deterministic=True)):7.2f} us")
for layout, n in (("cramped_room", 1000), ("cramped_room", 4096), ("asymmetric_advantages", 8192), ("cramped_room", 65536)):
    eng = OvercookedEngine(get_base_layout_params(layout, 200), n, 0)
    a = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
    eng.rollout(a, 200)
    us = min(eng.rollout_timed(a, 1000) for _ in range(3)) * 1e3 / 1000
    print(f"env {layout} x {n}: {us:6.2f} us per step")
    eng.close()

#!/usr/bin/env python3
"""Soak of the cooperative blocks by LDS hand-shake words (eight-wavefront policy kernel, 5x4 grid; round 6): a few hundred launches whose
tables cut the rows into stretches of every length 9 .. 41 blocks -- one tile per workgroup, and short tables split evenly so that a
workgroup's stretch runs through several tiles and networks (several shared blocks per launch and workgroup) -- each against the same
launch with every block on one wavefront (OC_POLICY_COOP=0): equal to f32 rounding, never a hang (run it under `timeout`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
from bench_fused_policy import random_net

W, H = 5, 4
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, gen) for _ in range(5)])
rng = np.random.default_rng(1)
os.environ["OC_POLICY_DUO"] = "0"
worst, shared, t0 = 0.0, 0, time.perf_counter()
for it in range(launches):
    if it % 3 == 2:                                                  # a short table, split evenly over 256 workgroups: stretches through several tiles
        nseg = int(rng.integers(3, 40))
        seg = int(rng.integers(2000, 9000))
        rows, rpt = nseg * seg, 1 << 20
        segments = [(g * seg, seg, int(rng.integers(0, 5)), TILE_ACT) for g in range(nseg)]
    else:                                                            # one tile per workgroup, `blocks` blocks each (the last tile ragged)
        blocks = int(rng.integers(9, 42))
        wgs = int(rng.integers(128, 257))
        rows, rpt = wgs * blocks * 32 - int(rng.integers(0, 31)), blocks * 32
        segments = [(0, rows, int(rng.integers(0, 5)), TILE_ACT)]
    obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.12).to(torch.int8)
    plan = PolicyPlan(nets, rows, segments, rows_per_tile=rpt)
    out = {}
    for coop in ("1", "0"):
        os.environ["OC_POLICY_COOP"] = coop
        logp = torch.full((rows, 6), float("nan"), device="cuda")
        act = torch.full((rows,), -1, dtype=torch.int32, device="cuda")
        plan.forward(obs, actions=act, logp=logp, deterministic=True)
        torch.cuda.synchronize()
        out[coop] = (logp, act)
    d = float((out["1"][0] - out["0"][0]).abs().max())
    assert d < 4e-6 and not bool(torch.isnan(out["1"][0]).any()) and int((out["1"][1] < 0).sum()) == 0, (it, d)
    worst = max(worst, d)
    shared += int(d > 0)
    if it % 50 == 49:
        print(f"{it + 1} launches, {shared} with shared blocks, largest difference {worst:.2e}, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"ok: {launches} launches, {shared} of them with shared blocks, largest |log-prob difference| {worst:.2e}")

#!/usr/bin/env python3
"""hipEvent time of ONE fused policy launch (product library) at the shapes the round's targets are quoted on: the one-block kernel with
its plain position loop (OC_POLICY_MONO=0; only differs where one wavefront runs per SIMD: every grid but 5x4), without shared blocks
(OC_POLICY_COOP=0), as it is, the two-blocks-per-wavefront kernel (OC_POLICY_DUO=1) and the default choice; best and median of several batches.

    python tools/time_policy_launch.py [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_fused_policy import random_net

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 4)
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, gen) for _ in range(8)])


def timed(fn, iters=100, batches=7):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    return min(out), float(np.median(out))


shapes = [("2 x 1000 (eval_1000, both agents)", 1000, 2, None), ("1 x 1000 (eval_1000, one agent)", 1000, 1, None), ("32 x 4096 (16 blocks per workgroup)", 4096, 32, None), ("15 x 4369 x 2 sides (bench line)", 4369, 30, None), ("15 x 4352 x 2", 4352, 30, None), ("245 760 rows, 1024 per tile", 61440, 4, 1024),
          ("8184 x 2 (config 4 share)", 264, 62, None)]
ref = {}
for name, seg, nseg, rpt in shapes:
    rows = seg * nseg
    obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
    plan = PolicyPlan(nets, rows, [(g * seg, seg, g % 8, TILE_ACT) for g in range(nseg)], rows_per_tile=rpt)
    actions = torch.zeros(rows, dtype=torch.int32, device="cuda")
    logp = torch.zeros((rows, 6), device="cuda")
    line = f"{name:36s} {rows:7d} rows {plan.n_tiles:4d} tiles:"
    modes = [("one block, plain loop, no coop", {"OC_POLICY_DUO": "0", "OC_POLICY_MONO": "0", "OC_POLICY_COOP": "0"}), ("one block, no coop", {"OC_POLICY_DUO": "0", "OC_POLICY_COOP": "0"}),
             ("one block", {"OC_POLICY_DUO": "0"}), ("one block, single blocks unshared", {"OC_POLICY_DUO": "0", "OC_POLICY_COOP": "4"}), ("one block, pairs share two leftover blocks", {"OC_POLICY_DUO": "0", "OC_POLICY_COOP": "2"}), ("duo", {"OC_POLICY_DUO": "1"}), ("default", {})]
    for label, env in modes:
        for k in ("OC_POLICY_DUO", "OC_POLICY_COOP", "OC_POLICY_MONO"):
            os.environ.pop(k, None)
        os.environ.update(env)
        best, med = timed(lambda: plan.forward(obs, actions=actions, logp=logp, deterministic=True))
        line += f"  {label} {best:6.2f} / {med:6.2f}"
        a, l = actions.clone(), logp.clone()
        if name not in ref:
            ref[name] = (a, l)
        else:
            # (bit for bit unless the one-block kernel shared a stretch's last blocks among wavefronts: those sum fc1 in another order)
            if label == "one block, no coop":                       # (the pipelined position loop: the same instructions in the same order)
                assert torch.equal(a, ref[name][0]) and torch.equal(l, ref[name][1]), "the pipelined position loop is not bit-identical"
            d = (l - ref[name][1]).abs().max().item()
            assert d < 4e-6 and (a != ref[name][0]).float().mean().item() < 1e-4, f"the kernels disagree: {d}"
    print(line + "  us (best / median of 7 x 100 launches)", flush=True)

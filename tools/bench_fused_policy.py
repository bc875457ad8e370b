#!/usr/bin/env python3
"""oc_policy_forward (fused MFMA policy inference) on the config-3-like rollout shape: cramped_room, 4096 kitchens
per pairing segment, 8 conventions (15 segments), actors + critics for both sides in ONE launch, next to the
batched-GEMM PyTorch path (StackedActors + StackedCritics) it replaces.  Random weights of the reference shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, CRITIC_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
from torch_policies import StackedActors, StackedCritics
from diverse_conventions_amd.sharding import segment_policy_indices
from diverse_conventions_amd.sharding import pairing_segments


def random_net(W, H, n_out, head, gen):
    P = (W - 2) * (H - 2)
    r = lambda *s: torch.randn(*s, generator=gen) * 0.1
    return {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
            "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
            head + ".weight": r(n_out, 64), head + ".bias": r(n_out)}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def run(layout, threads, k, digits=3, rows_per_tile=None, iters=50, with_torch=True):
    params = get_base_layout_params(layout, 200)
    W, H = params["width"], params["height"]
    segs = pairing_segments(threads, k)
    G = len(segs)
    n = threads * G
    gen = torch.Generator().manual_seed(0)
    actors = [random_net(W, H, 6, ACTOR_HEAD, gen) for _ in range(k + 1)]
    critics = [random_net(W, H, 1, CRITIC_HEAD, gen) for _ in range(2 * k + 1)]       # sp, xp0[i], xp1[i]
    eng = OvercookedEngine(params, n, 0)
    acts = torch.randint(0, 6, (2, n), dtype=torch.int32, device="cuda")
    for _ in range(30):
        eng.step(acts)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=critics, digits=digits)
    ego, partner = segment_policy_indices(segs, "ego"), segment_policy_indices(segs, "partner")
    segments = []
    for g in range(G):
        for side, pol in ((0, ego[g]), (1, partner[g])):
            row0 = side * n + g * threads
            segments.append((row0, threads, pol, TILE_ACT))
            segments.append((row0, threads, k + 1 + g, 0))                           # one critic per segment
    plan = PolicyPlan(nets, 2 * n, segments, rows_per_tile=rows_per_tile)
    actions = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    alp = torch.zeros(2 * n, device="cuda")
    values = torch.zeros(2 * n, device="cuda")
    fused = lambda: plan.forward(eng.obs, raw=values, actions=actions, action_logp=alp)
    us = timed(fused, iters)
    P = (W - 2) * (H - 2)
    rows = 2 * 2 * n                                                                 # actor + critic evaluations
    flop = rows * 2 * (P * 180 * 32 + P * 32 * 64 + 64 * 64 + 64 * (6 + 1) / 2)
    mfma_cycles = rows / 32 * 32 * (P * (6 * digits + 24) + 48 + 24)      # i8 conv digits + bf16 piece products, 32 cycles each
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    out = {"layout": layout, "kitchens": n, "segments": G, "conventions": k + 1, "digits": digits, "tiles": plan.n_tiles,
           "rows_per_tile": plan.rows_per_tile, "network_evaluations": rows, "fused_us": round(us, 1),
           "useful_TFLOPs": round(flop / us * 1e-6, 1),
           "mfma_issue_bound_us_at_2.4GHz": round(mfma_cycles / (cu * 4) / 2400, 1),
           "evals_per_s": round(rows / us * 1e6)}
    if with_torch:
        sa = StackedActors(actors, (W, H, 20), "cuda")
        sc = StackedCritics(critics, (W, H, 20), "cuda")
        sel_a = [sa.select(torch.tensor(ix)) for ix in (ego, partner)]
        sel_c = sc.select(torch.arange(G))
        def torch_path():
            for p in range(2):
                obs = eng.obs[p].view(G, threads, W, H, 20)
                sa.act(obs, selected=sel_a[p])
                sc.values(obs, selected=sel_c)
        out["torch_batched_gemm_us"] = round(timed(torch_path, 20), 1)
        # agreement of the two paths on the same observations (deterministic actions)
        plan.forward(eng.obs, raw=values, actions=actions, action_logp=alp, deterministic=True)
        a_t = torch.stack([sa.act(eng.obs[p].view(G, threads, W, H, 20), deterministic=True, selected=sel_a[p])[0].view(-1) for p in range(2)])
        v_t = torch.stack([sc.values(eng.obs[p].view(G, threads, W, H, 20), selected=sel_c).view(-1) for p in range(2)])
        out["actions_equal_frac"] = float((a_t.view(-1) == actions).float().mean())
        out["values_max_abs_diff"] = float((v_t.view(-1) - values).abs().max())
    eng.close()
    return out


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    print(json.dumps(run("cramped_room", 4096, 7)), flush=True)
    if not quick:
        print(json.dumps(run("cramped_room", 4096, 7, digits=3, with_torch=False)), flush=True)
        for rpt in (128, 256, 512, 1024):
            print(json.dumps(run("cramped_room", 4096, 7, rows_per_tile=rpt, with_torch=False)), flush=True)
        print(json.dumps(run("cramped_room", 546, 7)), flush=True)
        print(json.dumps(run("asymmetric_advantages", 1024, 7)), flush=True)

#!/usr/bin/env python3
"""Mixed-play collection at scale: L = 200, kitchens = 64 x 199 = 12 736 (and 65 536 / 199 -> 329 x 199),
2L steps, per step one fused policy launch in which the trained policy's and the partner convention's actor and
critic cover the same rows (per-row select; blocks of kitchens already forced onto the trained policy are skipped by
the partner networks) + one env launch; then the reference-layout buffer as one gather."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_paired_rollout import random_actor, random_critic
from diverse_conventions_amd import MixedPlayRollout, OvercookedEngine, PackedNetworks, get_base_layout_params


def run(layout, reps, L=200):
    params = get_base_layout_params(layout, L)
    W, H = params["width"], params["height"]
    n = reps * (L - 1)
    gen = torch.Generator().manual_seed(0)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_actor(W, H, gen=gen) for _ in range(2)],
                                           critics=[random_critic(W, H, gen) for _ in range(2)])
    eng = OvercookedEngine(params, n, 0)
    ro = MixedPlayRollout(eng, nets, L, main=(0, 2), partner=(1, 3))
    ro.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro.collect()
    torch.cuda.synchronize(); t_collect = time.perf_counter() - t0
    ro.buffer()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf = ro.buffer()
    torch.cuda.synchronize(); t_buf = time.perf_counter() - t0
    out = {"layout": layout, "kitchens": n, "L": L, "collect_ms": round(t_collect * 1e3, 2), "us_per_step": round(t_collect / (2 * L) * 1e6, 1),
           "agent_env_steps_per_s": round(2 * n * 2 * L / t_collect), "buffer_gather_ms": round(t_buf * 1e3, 3),
           "kept_steps_by_partner_policy": int(buf["choice"].abs().sum())}
    eng.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run("cramped_room", 64)), flush=True)
    print(json.dumps(run("cramped_room", 329)), flush=True)
    print(json.dumps(run("asymmetric_advantages", 64)), flush=True)

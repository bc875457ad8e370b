#!/usr/bin/env python3
"""How the pieces compose (not a trainer): one PPO-style data-collection segment on the GPU --
paired rollout (batched actors on both sides + fused env step) written straight into an int8 time-major
trajectory, batched critics over the whole trajectory, one-launch GAE -- with the timings of each part."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_paired_rollout import random_actor
from diverse_conventions_amd import (OvercookedEngine, StackedActors, StackedCritics, TrajectoryBuffer, gae_returns,
                                     get_base_layout_params, pairing_segments)
from diverse_conventions_amd.policy import segment_policy_indices


def main(layout="cramped_room", threads=1024, k=3, T=200):
    params = get_base_layout_params(layout, T)
    W, H = params["width"], params["height"]
    segs = pairing_segments(threads, k)
    n = threads * len(segs)
    gen = torch.Generator().manual_seed(0)
    actors = StackedActors([random_actor(W, H, gen=gen) for _ in range(k + 1)], (W, H, 20), "cuda")
    csd = []
    for _ in range(len(segs)):                              # one critic per segment (sp, xp0_i, xp1_i)
        sd = random_actor(W, H, gen=gen)
        sd["v_out.weight"], sd["v_out.bias"] = torch.randn(1, 64, generator=gen) * 0.1, torch.zeros(1)
        csd.append(sd)
    critics = StackedCritics(csd, (W, H, 20), "cuda")
    eng = OvercookedEngine(params, n, 0)
    buf = TrajectoryBuffer(eng, T)
    sel = [actors.select(torch.tensor(segment_policy_indices(segs, side))) for side in ("ego", "partner")]
    G = len(segs)

    torch.cuda.synchronize(); t0 = time.perf_counter()
    obs = buf.start()
    for t in range(T):
        acts = [actors.act(obs[p].view(G, threads, W, H, 20), selected=sel[p])[0].view(-1) for p in range(2)]
        obs, rew, done = buf.step(torch.stack(acts))
    torch.cuda.synchronize(); t_roll = time.perf_counter() - t0

    t0 = time.perf_counter()
    csel = critics.select(torch.arange(G))
    ego_obs = buf.agent_view(0)                              # [T+1, N, W, H, C] int8, no copy
    values = torch.stack([critics.values(ego_obs[t].view(G, threads, W, H, 20), selected=csel).view(n, 1, 1)
                          for t in range(T + 1)])            # [T+1, N, 1, 1]
    torch.cuda.synchronize(); t_val = time.perf_counter() - t0

    t0 = time.perf_counter()
    rewards = buf.reward[:, 0].float().view(T, n, 1, 1)
    masks = torch.ones((T + 1, n, 1, 1), device="cuda")
    masks[1:] = (1 - buf.done.float()).view(T, n, 1, 1)
    returns, adv = gae_returns(rewards, values.contiguous(), masks, gamma=0.99, gae_lambda=0.95)
    torch.cuda.synchronize(); t_gae = time.perf_counter() - t0
    print(json.dumps({"layout": layout, "kitchens": n, "segments": G, "T": T,
                      "rollout_s": round(t_roll, 4), "agent_env_steps_per_s": round(2 * n * T / t_roll),
                      "critic_pass_s": round(t_val, 4), "gae_s": round(t_gae, 5),
                      "mean_return_per_kitchen": round(float(buf.reward[:, 0].sum()) / n, 3),
                      "adv_std": round(float(adv.std()), 4), "trajectory_obs_GB": round(buf.obs.numel() / 1e9, 2)}))
    eng.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How the pieces compose (not a trainer): one PPO-style data-collection segment on the GPU and the update-side data
path behind it, with the time of each part --
  rollout      FusedPairedRollout.collect: per step ONE fused MFMA policy launch (every convention's actor and the
               segment's critic, both sides) + ONE env launch writing the next int8 trajectory slot (oc_policy_rollout)
  bootstrap    one more policy launch for the critics' values of the last observation
  GAE          oc_gae_returns over the whole [T, N, 2] batch in one launch
  normalise    oc_advantage_normalize
  minibatches  one epoch of MinibatchSource.feed_forward_generator (device permutation, int8 row gather, packed scalars)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_paired_rollout import random_actor, random_critic
from diverse_conventions_amd import (FusedPairedRollout, MinibatchSource, OvercookedEngine, PackedNetworks, PolicyPlan,
                                     TrajectoryBuffer, gae_returns, get_base_layout_params, normalize_advantages)


def clock(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def main(layout="cramped_room", threads=2048, k=3, T=200, num_mini_batch=4):
    params = get_base_layout_params(layout, T)
    W, H = params["width"], params["height"]
    G = 2 * k + 1
    n = threads * G
    gen = torch.Generator().manual_seed(0)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_actor(W, H, gen=gen) for _ in range(k + 1)],
                                           critics=[random_critic(W, H, gen) for _ in range(G)])
    critic_of = [[k + 1 + g for g in range(G)]] * 2                 # one critic per segment (sp, xp0_i, xp1_i), both sides
    eng = OvercookedEngine(params, n, 0)
    ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=critic_of)
    traj = TrajectoryBuffer(eng, T)
    logp = torch.zeros((T, 2, n), device="cuda")
    values = torch.zeros((T + 1, 2, n), device="cuda")
    ro.collect(traj, logp=logp, values=values[:T])                  # warm-up (JIT, first touches)

    _, t_roll = clock(lambda: ro.collect(traj, logp=logp, values=values[:T]))
    boot = PolicyPlan(nets, 2 * n, [(side * n + g * threads, threads, critic_of[side][g], 0) for side in range(2) for g in range(G)])
    _, t_boot = clock(lambda: boot.forward(traj.obs[T], raw=values[T]))

    def gae():
        # reference layout [T(+1), N, A, 1]: agent-major engine tensors transposed once (small: 4 B per sample)
        to_ref = lambda t: t.permute(0, 2, 1).unsqueeze(-1).contiguous()
        rewards = to_ref(traj.reward.float())
        masks = torch.ones((T + 1, n, 2, 1), device="cuda")
        masks[1:] = (1 - traj.done.float()).view(T, n, 1, 1)
        vals = to_ref(values)
        returns, adv = gae_returns(rewards, vals, masks, gamma=0.99, gae_lambda=0.95)
        return rewards, masks, vals, returns, adv
    gae()                                                           # warm-up: allocator, first launches
    (rewards, masks, vals, returns, adv), t_gae = clock(gae)
    normalize_advantages(adv, std_over="active")
    (adv_n, stats), t_norm = clock(lambda: normalize_advantages(adv, std_over="active"))

    src = MinibatchSource.from_trajectory(traj, vals, returns, masks, logp.permute(0, 2, 1).unsqueeze(-1).contiguous())

    def epoch():
        seen = 0
        for batch in src.feed_forward_generator(adv_n, num_mini_batch=num_mini_batch):
            seen += batch[1].shape[0]
        return seen
    epoch()
    seen, t_epoch = clock(epoch)
    print(json.dumps({"layout": layout, "kitchens": n, "segments": G, "conventions": k + 1, "T": T,
                      "rollout_ms": round(t_roll * 1e3, 2), "agent_env_steps_per_s": round(2 * n * T / t_roll),
                      "bootstrap_values_ms": round(t_boot * 1e3, 3), "gae_ms": round(t_gae * 1e3, 3),
                      "advantage_normalise_ms": round(t_norm * 1e3, 3), "minibatch_epoch_ms": round(t_epoch * 1e3, 2),
                      "samples_per_epoch": seen, "mean_return_per_kitchen": round(float(traj.reward[:, 0].sum()) / n, 3),
                      "trajectory_obs_GB": round(traj.obs.numel() / 1e9, 2)}))
    eng.close()


if __name__ == "__main__":
    main()
    main("asymmetric_advantages", threads=512)

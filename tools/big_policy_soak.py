#!/usr/bin/env python3
"""Full-size soak of the rollout with policies in the loop: 8 conventions in 15 pairing segments of 4096 kitchens
(61 440 kitchens), 1000 steps of fused policy launch + env launch (5 horizons), collected in segments of 100 steps into
a time-major trajectory.  Every step of a sample of kitchens is checked: the env against the oracle driven by the
recorded actions, and -- every 50th step -- the recorded actions / log-probs against the PyTorch batched evaluation of
the same reference-format weights on the recorded observations (deterministic draws)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bench_paired_rollout import random_actor
from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, TrajectoryBuffer, get_base_layout_params
from torch_policies import StackedActors
from diverse_conventions_amd.sharding import segment_policy_indices
from oracle import oracle

threads, k, T, rounds = 4096, 7, 100, 10
params = get_base_layout_params("cramped_room", 200)
G, n = 2 * k + 1, threads * (2 * k + 1)
gen = torch.Generator().manual_seed(0)
sds = [random_actor(5, 4, gen=gen) for _ in range(k + 1)]
nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=sds)
sa = StackedActors(sds, (5, 4, 20), "cuda")
eng = OvercookedEngine(params, n, 0)
ro = FusedPairedRollout(eng, nets, threads, k, deterministic=True)
traj = TrajectoryBuffer(eng, T)
logp = torch.zeros((T, 2, n), device="cuda")
rng = np.random.default_rng(0)
sample = np.sort(rng.choice(n, 256, replace=False)); sample[0], sample[-1] = 0, n - 1
idx = torch.from_numpy(sample).cuda()
ora = oracle.OracleVecEnv(params, len(sample))
sel = [sa.select(torch.tensor(segment_policy_indices(ro.segments, side))) for side in ("ego", "partner")]
ok_env = ok_pol = True
dones = 0
t0 = time.perf_counter()
gpu_s = 0.0
for r in range(rounds):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ro.collect(traj, logp=logp)
    torch.cuda.synchronize(); gpu_s += time.perf_counter() - t1
    acts = traj.actions[:, :, idx].cpu().numpy()
    obs_s, rew_s, done_s = traj.obs[:, :, idx].cpu().numpy(), traj.reward[:, :, idx].cpu().numpy(), traj.done[:, idx].cpu().numpy()
    if r == 0:
        ok_env &= np.array_equal(obs_s[0], ora.encode())
    for t in range(T):
        oobs, orew, odone = ora.step(acts[t])
        ok_env &= np.array_equal(obs_s[t + 1], oobs) and np.array_equal(rew_s[t], orew) and np.array_equal(done_s[t], odone)
        dones += int(odone.sum())
    for t in (0, 50, 99):
        for p in range(2):
            a, lp = sa.act(traj.obs[t, p].view(G, threads, 5, 4, 20), deterministic=True, selected=sel[p])
            ok_pol &= bool((a.view(-1) == traj.actions[t, p]).all()) and float((lp.view(-1) - logp[t, p]).abs().max()) < 4e-6
    traj.after_update()
print(json.dumps({"kitchens": n, "steps": rounds * T, "sampled_kitchens": len(sample), "env_bit_exact_on_sample": bool(ok_env),
                  "policy_matches_pytorch_on_checked_steps": bool(ok_pol), "episode_ends_seen_in_sample": dones,
                  "rollout_us_per_step": round(gpu_s / (rounds * T) * 1e6, 1),
                  "agent_env_steps_per_s": round(2 * n * rounds * T / gpu_s), "wall_s_with_checks": round(time.perf_counter() - t0, 1)}))
eng.close()

#!/usr/bin/env python3
"""Large-batch sanity: a million-plus kitchens (ragged count), sampled against the oracle, plus timing."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import algorithmic_bytes
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from oracle import oracle

for layout, n in (("cramped_room", 1_000_003), ("asymmetric_advantages", 524_289)):
    params = get_base_layout_params(layout, 200)
    eng = OvercookedEngine(params, n, 0)
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(n, 300, replace=False)); sample[-1] = n - 1; sample[0] = 0
    ora = oracle.OracleVecEnv(params, len(sample))
    idx = torch.from_numpy(sample).cuda()
    ok = True
    for t in range(205):
        a = torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32)
        obs, rew, done = eng.step(a)
        oobs, orew, odone = ora.step(a[:, idx].cpu().numpy())
        ok &= np.array_equal(obs[:, idx].cpu().numpy(), oobs) and np.array_equal(rew[:, idx].cpu().numpy(), orew)
        ok &= int(done.sum()) == (n if t == 199 else 0)
    acts = torch.randint(0, 6, (8, 2, n), device="cuda", dtype=torch.int32)
    ms = min(eng.rollout_timed(acts, 50) for _ in range(2)) / 50
    S = params["width"] * params["height"]
    print(json.dumps({"layout": layout, "kitchens": n, "bit_exact_on_sample": bool(ok), "ms_per_launch": round(ms, 4),
                      "agent_env_steps_per_s": round(2 * n / ms * 1e3), "roofline_frac": round(algorithmic_bytes(S) * n / ms / 1e6 / 8000, 3)}))
    eng.close()

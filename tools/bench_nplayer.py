#!/usr/bin/env python3
"""us per step of the N-player group kernel (csrc/oc_np_kernels.h; rounds 2-4: a plain kernel pair): multiplayer_schelling with 4 and 3 players,
simple_single, at 65 536 and 8192 kitchens, outputs in place.  Bytes per kitchen-step: P*S*(5P+10) obs + 2*record + 4P
actions + 4P reward + 4 done."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

for layout, maxp in (("multiplayer_schelling", None), ("multiplayer_schelling", 3), ("simple_single", None)):
    params = get_base_layout_params(layout, 200, max_num_players=maxp)
    P, S = params["num_players"], params["width"] * params["height"]
    for n in (65536, 8192):
        eng = OvercookedEngine(params, n, 0)
        acts = torch.randint(0, 6, (64, P, n), device="cuda", dtype=torch.int32)
        eng.rollout(acts, 50)
        us = min(eng.rollout_timed(acts, 300) for _ in range(3)) * 1e3 / 300
        nbytes = n * (P * S * (5 * P + 10) + 2 * eng.record_bytes + 8 * P + 4)
        print(json.dumps({"layout": layout, "players": P, "kitchens": n, "us_per_step": round(us, 1), "agent_env_steps_per_s": round(P * n / us * 1e6),
                          "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3), "kernel": eng.kernel_build_info}), flush=True)
        eng.close()

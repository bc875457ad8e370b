#!/usr/bin/env python3
"""Roofline fraction of the step kernel vs kitchens per GPU (default launch geometry), one launch per step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import algorithmic_bytes
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

for layout in ("cramped_room", "asymmetric_advantages"):
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    for n in (4096, 16384, 65536, 131072, 262144, 1048576):
        eng = OvercookedEngine(params, n, 0)
        acts = torch.randint(0, 6, (16, 2, n), device="cuda", dtype=torch.int32)
        steps = max(50, min(1000, 30_000_000 // n))
        eng.rollout(acts, 30)
        us = min(eng.rollout_timed(acts, steps) for _ in range(3)) * 1e3 / steps
        print(json.dumps({"layout": layout, "kitchens": n, "us_per_launch": round(us, 2),
                          "agent_env_steps_per_s": round(2 * n / us * 1e6),
                          "roofline_frac": round(algorithmic_bytes(S) * n / us / 1e3 / 8000, 3),
                          "obs_MB": round(2 * n * S * 20 / 1e6, 1)}), flush=True)
        eng.close()

#!/usr/bin/env python3
"""Does the N-player step's time depend on how the viewer planes lie relative to each other (channel aliasing of P store streams at a
power-of-two-ish stride)?  4-player multiplayer_schelling at kitchen counts around 65 536, and repeated engine creation at 65 536."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
params = get_base_layout_params("multiplayer_schelling", 200)
P, S = 4, params["width"] * params["height"]
def run(n, keep=None):
    acts = torch.randint(0, 6, (64, P, n), device="cuda", dtype=torch.int32)
    eng = OvercookedEngine(params, n, 0)
    eng.rollout(acts, 60)
    us = [round(eng.rollout_timed(acts, 300) * 1e3 / 300, 1) for _ in range(3)]
    nbytes = n * (P * S * 30 + 2 * eng.record_bytes + 8 * P + 4)
    print(json.dumps({"kitchens": n, "us": us, "frac": round(nbytes / min(us) / 1e3 / 8000, 3), "obs_ptr_mod_2MB": eng.obs.data_ptr() % (2 << 20),
                      "plane_stride_mod_64KB": (n * S * 30) % 65536}), flush=True)
    if keep is not None:
        keep.append(eng)
    else:
        eng.close()
for n in (65536, 65536, 65552, 65600, 66560, 61440, 70000, 65536):
    run(n)
keep = []
for _ in range(3):
    run(65536, keep)

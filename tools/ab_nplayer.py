#!/usr/bin/env python3
"""A/B of N-player kernel variants on one box, interleaved: OC_DIAGNOSTICS=1 python tools/ab_nplayer.py
Variants are environment settings read per launch by the diagnostics build (OC_NP_NOPIPE, OC_NP_SM) and tunings."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

VARIANTS = [("pipe", {}, None), ("nopipe", {"OC_NP_NOPIPE": "1"}, None)]
for P, n in ((4, 65536), (4, 8192), (3, 65536), (1, 65536), (1, 8192)):
    layout, maxp = ("simple_single", None) if P == 1 else ("multiplayer_schelling", P)
    params = get_base_layout_params(layout, 200, max_num_players=maxp)
    acts = torch.randint(0, 6, (64, P, n), device="cuda", dtype=torch.int32)
    eng = OvercookedEngine(params, n, 0)
    eng.rollout(acts, 50)
    res = {name: [] for name, _, _ in VARIANTS}
    for rep in range(4):
        for name, env, tune in VARIANTS:
            for k in ("OC_NP_NOPIPE", "OC_NP_SM"):
                os.environ.pop(k, None)
            os.environ.update(env)
            eng.rollout(acts, 20)
            res[name].append(round(eng.rollout_timed(acts, 300) * 1e3 / 300, 2))
    print(json.dumps({"players": P, "kitchens": n, **res}), flush=True)
    eng.close()

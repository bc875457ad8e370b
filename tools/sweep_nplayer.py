#!/usr/bin/env python3
"""Geometry / store-mode sweep of the N-player group kernel (csrc/oc_np_kernels.h): us per step, outputs in place.
    OC_DIAGNOSTICS=1 python tools/sweep_nplayer.py [players ...]      (the store mode knob OC_NP_SM exists in the diagnostics build only)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

players = [int(x) for x in sys.argv[1:]] or [4, 3, 1]
for P in players:
    layout, maxp = ("simple_single", None) if P == 1 else ("multiplayer_schelling", P)
    params = get_base_layout_params(layout, 200, max_num_players=maxp)
    S = params["width"] * params["height"]
    for n in [int(x) for x in os.environ.get("OC_SWEEP_N", "65536,8192").split(",")]:
        acts = torch.randint(0, 6, (64, P, n), device="cuda", dtype=torch.int32)
        best = None
        for sm in [int(x) for x in os.environ.get("OC_SWEEP_SM", "0,2,4").split(",")]:
            os.environ["OC_NP_SM"] = str(sm)
            for K, G, w in ((0, 0, 0), (16, 2, 8), (16, 1, 8), (16, 3, 8), (16, 4, 6), (8, 2, 8), (8, 2, 12), (8, 4, 8), (16, 2, 6), (16, 2, 4), (4, 2, 16), (12, 2, 8), (32, 2, 8), (32, 3, 8), (32, 4, 8), (40, 3, 8), (20, 2, 8)):
                if K * P > 64:
                    continue
                if sm == 4 and (K, G, w) not in ((0, 0, 0), (16, 2, 8)):
                    continue
                eng = OvercookedEngine(params, n, 0)
                try:
                    if K:
                        eng.set_tuning(K, G, w)
                    eng.rollout(acts, 30)
                    us = min(eng.rollout_timed(acts, 200) for _ in range(5)) * 1e3 / 200
                except Exception as exc:                              # noqa: BLE001
                    print(json.dumps({"players": P, "kitchens": n, "sm": sm, "K": K, "G": G, "wpc": w, "error": str(exc)[:80]}), flush=True)
                    eng.close()
                    continue
                nbytes = n * (P * S * (5 * P + 10) + 2 * eng.record_bytes + 8 * P + 4)
                row = {"players": P, "kitchens": n, "sm": sm, "K": K, "G": G, "wpc": w, "us_per_step": round(us, 2), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3)}
                print(json.dumps(row), flush=True)
                if best is None or us < best["us_per_step"]:
                    best = row
                eng.close()
        print("BEST", json.dumps(best), flush=True)

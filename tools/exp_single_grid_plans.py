#!/usr/bin/env python3
"""Config 4's per-GPU share (asymmetric_advantages, 8184 kitchens, 16 conventions = 31 segments of 264 kitchens, actors + critics)
with the policies in the loop: the single-grid plan of FusedPairedRollout against the multi-grid planner run on one grid
(GroupPairedRollout over a one-engine group: packed stretches / wave tiles), OC_POLICY_WAVE_TILES=0/1 forcing either."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import EngineGroup, FusedPairedRollout, GroupPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params

layout, n, k, T = (sys.argv[1] if len(sys.argv) > 1 else "asymmetric_advantages"), int(sys.argv[2]) if len(sys.argv) > 2 else 8192, 15, 300
G = 2 * k + 1
threads = n // G
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
P = (W - 2) * (H - 2)
gen = torch.Generator().manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=gen) * 0.1
trunk = lambda: {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                 "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64)}
actors = [dict(trunk(), **{"act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}) for _ in range(k + 1)]
crit = [dict(trunk(), **{"v_out.weight": r(1, 64), "v_out.bias": r(1)}) for _ in range(G)]
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=crit)
cn = [[k + 1 + g for g in range(G)]] * 2


def timed(step):
    step(30)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        step(T)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / T * 1e6)
    return round(best, 1)


eng = OvercookedEngine(params, threads * G, 0)
ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=cn)
print(json.dumps({"layout": layout, "kitchens": threads * G, "plan": "FusedPairedRollout (single-grid plan, one tile per workgroup)", "us_per_step": timed(ro.step)}), flush=True)
eng.close()
for wt in ("0", "1"):
    os.environ["OC_POLICY_WAVE_TILES"] = wt
    eng = OvercookedEngine(params, threads * G, 0)
    grp = EngineGroup([eng])
    g = GroupPairedRollout(grp, [nets], threads, k, critic_nets=[cn], partition="one_launch")
    print(json.dumps({"layout": layout, "plan": f"multi-grid planner on one grid, OC_POLICY_WAVE_TILES={wt}: {g.multi_plan.tiles_per_workgroup} tiles per workgroup, "
                      f"{g.multi_plan.waves_per_workgroup} wavefront(s) per workgroup", "us_per_step": timed(g.step)}), flush=True)
    g.close()
    grp.close(close_engines=True)

#!/usr/bin/env python3
"""What the action draw costs inside the fused policy launch: the bench line's shape (8 conventions, 15 segments of 4369 kitchens, actors
on both sides: 131 070 evaluations) with sampled actions against the same launch with argmax actions (no hash, no noise)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params

def run(layout="cramped_room", n_total=65536, k=7, iters=300):
    params = get_base_layout_params(layout, 200)
    W, H = params["width"], params["height"]
    P = (W - 2) * (H - 2)
    g = torch.Generator().manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=g) * 0.1
    sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P), "base.cnn.cnn.3.bias": r(64),
            "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64), "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(k + 1)]
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
    threads = n_total // (2 * k + 1)
    eng = OvercookedEngine(params, threads * (2 * k + 1), 0)
    out = {"layout": layout, "evaluations": 2 * eng.num_kitchens}
    for det in (False, True):
        ro = FusedPairedRollout(eng, nets, threads, k, deterministic=det)
        ro.step(50)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(iters):
                ro.plan.forward(eng.obs, actions=ro.actions, action_logp=ro.logp, deterministic=det, seed=0, counter=0, device_counter=ro.draw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        out["argmax_us" if det else "sampled_us"] = round(best, 2)
    eng.close()
    return out

if __name__ == "__main__":
    for layout in (sys.argv[1:] or ["cramped_room", "asymmetric_advantages"]):
        print(json.dumps(run(layout)), flush=True)

#!/usr/bin/env python3
"""Policies in the loop, stepping in place against collecting into time-major slots (what training does), and the store mode of the
fresh slots: write-through non-temporal (the default for fresh destinations, best for the env launch alone) against plain
write-through -- the NEXT launch is the policy kernel, which reads those very planes.  Run once per OC_STORE_MODE (read at
oc_create): OC_STORE_MODE=2,4 (default) / 2,2 / 2,1 / 2,0."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, FusedPairedRollout, PackedNetworks, TrajectoryBuffer, get_base_layout_params
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_paired_rollout import random_actor, random_critic

def run(layout, threads, k, critics, T=200):
    params = get_base_layout_params(layout, 200)
    W, H = params["width"], params["height"]
    G = 2 * k + 1
    gen = torch.Generator().manual_seed(0)
    actors = [random_actor(W, H, gen=gen) for _ in range(k + 1)]
    crit = [random_critic(W, H, gen) for _ in range(G)] if critics else []
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=crit)
    eng = OvercookedEngine(params, threads * G, 0)
    if os.environ.get("OC_EXP_TUNING"):                    # "kpw,passes,wpc": the env kernel's geometry for every destination kind
        eng.set_tuning(*[int(v) for v in os.environ["OC_EXP_TUNING"].split(",")])
    ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=[[k + 1 + g for g in range(G)]] * 2 if critics else None)
    ro.step(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); ro.step(T); torch.cuda.synchronize()
    in_place = (time.perf_counter() - t0) / T * 1e6
    traj = TrajectoryBuffer(eng, T)
    ro.collect(traj); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ro.collect(traj); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / T * 1e6)
    eng.close()
    return {"store_mode": os.environ.get("OC_STORE_MODE", "default"), "layout": layout, "kitchens": threads * G, "critics": critics,
            "in_place_us_per_step": round(in_place, 2), "collect_time_major_us_per_step": round(best, 2)}

print(json.dumps(run("cramped_room", 4369, 7, False)), flush=True)
print(json.dumps(run("asymmetric_advantages", 264, 15, True)), flush=True)

#!/usr/bin/env python3
"""Whole rollout step = policies (both sides, all pairing segments) + fused env step, eager vs one HIP graph replay
per step: the batched-GEMM PyTorch path (StackedActors) and the fused MFMA kernel (FusedPairedRollout, actors only
and actors + critics).  Random-weight networks of the reference architecture (hidden 64)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from torch_policies import StackedActors
from diverse_conventions_amd.rollout import FusedPairedRollout
from torch_policies import PairedRollout
from diverse_conventions_amd.fused_policy import PackedNetworks


def random_actor(W, H, hidden=64, gen=None):
    P = (W - 2) * (H - 2)
    r = lambda *s: torch.randn(*s, generator=gen) * 0.1
    return {"base.cnn.cnn.0.weight": r(hidden // 2, 20, 3, 3), "base.cnn.cnn.0.bias": r(hidden // 2),
            "base.cnn.cnn.3.weight": r(hidden, hidden // 2 * P), "base.cnn.cnn.3.bias": r(hidden),
            "base.cnn.cnn.5.weight": r(hidden, hidden), "base.cnn.cnn.5.bias": r(hidden),
            "act.action_out.linear.weight": r(6, hidden), "act.action_out.linear.bias": r(6)}


def random_critic(W, H, gen):
    sd = random_actor(W, H, gen=gen)
    sd["v_out.weight"], sd["v_out.bias"] = torch.randn(1, 64, generator=gen) * 0.1, torch.randn(1, generator=gen) * 0.1
    return sd


def time_steps(ro, steps):
    for _ in range(10):
        ro.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ro.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def run_fused(layout, threads, k, steps=200, critics=False):
    params = get_base_layout_params(layout, 200)
    W, H = params["width"], params["height"]
    n, G = threads * (2 * k + 1), 2 * k + 1
    gen = torch.Generator().manual_seed(0)
    actors = [random_actor(W, H, gen=gen) for _ in range(k + 1)]
    crit = [random_critic(W, H, gen) for _ in range(G)] if critics else []
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=crit)
    cn = [[k + 1 + g for g in range(G)]] * 2 if critics else None
    out = {"layout": layout, "kitchens": n, "segments": G, "conventions": k + 1, "policy": "fused MFMA kernel, f32-class numerics",
           "critics": critics}
    for mode in ("eager", "graph", "enqueued_from_c"):
        eng = OvercookedEngine(params, n, 0)
        ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=cn)
        if mode == "graph":
            ro.capture()
        if mode == "enqueued_from_c":                     # one call enqueues all steps (oc_policy_rollout)
            ro.step(10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ro.step(steps)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
        else:
            us = time_steps(ro, steps)
        out[f"{mode}_us_per_step"] = round(us, 1)
        out[f"{mode}_agent_env_steps_per_s"] = round(2 * n / us * 1e6)
        eng.close()
    return out


def run(layout, threads, k, steps=200, dtype=torch.float32):
    params = get_base_layout_params(layout, 200)
    n = threads * (2 * k + 1)
    gen = torch.Generator().manual_seed(0)
    actors = StackedActors([random_actor(params["width"], params["height"], gen=gen) for _ in range(k + 1)],
                           (params["width"], params["height"], 20), "cuda", dtype=dtype)
    out = {"layout": layout, "kitchens": n, "segments": 2 * k + 1, "conventions": k + 1, "policy_dtype": str(dtype)}
    for mode in ("eager", "graph"):
        eng = OvercookedEngine(params, n, 0)
        ro = PairedRollout(eng, actors, threads, k)
        if mode == "graph":
            ro.capture()
        for _ in range(10):
            ro.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ro.step()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / steps * 1e6
        out[f"{mode}_us_per_step"] = round(us, 1)
        out[f"{mode}_agent_env_steps_per_s"] = round(2 * n / us * 1e6)
        eng.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run_fused("cramped_room", 546, 7)), flush=True)
    print(json.dumps(run_fused("cramped_room", 4096, 7)), flush=True)
    print(json.dumps(run_fused("cramped_room", 4096, 7, critics=True)), flush=True)
    print(json.dumps(run_fused("asymmetric_advantages", 1024, 7, critics=True)), flush=True)
    # one GPU's share of BASELINE config 4: asymmetric_advantages, 65 536 kitchens over 8 GPUs, 16 conventions (31 segments)
    print(json.dumps(run_fused("asymmetric_advantages", 8192 // 31, 15, critics=True)), flush=True)
    print(json.dumps(run("cramped_room", 546, 7)), flush=True)          # config-3-sized: 8190 kitchens
    print(json.dumps(run("cramped_room", 4096, 7)), flush=True)         # 61 440 kitchens
    print(json.dumps(run("cramped_room", 4096, 7, dtype=torch.bfloat16)), flush=True)

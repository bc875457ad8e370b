#!/usr/bin/env python3
"""K-steps-per-launch rollout mode: the whole synthetic rollout in one launch, per-kitchen
state resident in LDS, actions from the counter-hash stream or a tape.  Bytes per kitchen-step that must move
here: obs + actions (tape mode) + reward + done; the state round trip of the one-launch-per-step path is gone."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params


def run(layout, n, T, mode, time_major):
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, n, 0)
    tape = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32) if mode == "tape" else None
    kw = {}
    if time_major:
        kw = dict(obs=torch.empty((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda"),
                  reward=torch.empty((T, 2, n), dtype=torch.int32, device="cuda"),
                  done=torch.empty((T, n), dtype=torch.int32, device="cuda"))
    eng.rollout_fused(T, actions=tape, seed=1, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); eng.rollout_fused(T, actions=tape, seed=1, **kw); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    us = best * 1e3 / T
    bytes_step = n * (2 * S * 20 + 8 + 4 + (8 if mode == "tape" else 0))
    out = {"mode": f"fused {T}-step launch, {mode} actions, " + ("time-major outputs" if time_major else "outputs in place"),
           "layout": layout, "kitchens": n, "us_per_step": round(us, 2), "agent_env_steps_per_s": round(2 * n / us * 1e6),
           "bytes_per_step_MB": round(bytes_step / 1e6, 1), "GBps": round(bytes_step / us / 1e3, 1), "roofline_frac": round(bytes_step / us / 1e3 / 8000, 3)}
    eng.close()
    return out


if __name__ == "__main__":
    for args in (("cramped_room", 65536, 200, "hash", False), ("cramped_room", 65536, 200, "tape", False),
                 ("cramped_room", 65536, 200, "tape", True), ("asymmetric_advantages", 65536, 100, "hash", False),
                 ("cramped_room", 4096, 200, "hash", False)):
        print(json.dumps(run(*args)), flush=True)

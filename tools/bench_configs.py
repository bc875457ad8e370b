#!/usr/bin/env python3
"""Secondary measurements on one GPU for the BASELINE.json configs that are not the headline bench line,
plus the time-major-destination variant of the headline (observations written to a fresh slot every step).
Prints one JSON object per measurement."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import algorithmic_bytes
from diverse_conventions_amd import CLASSIC_LAYOUTS, OvercookedEngine, get_base_layout_params, pairing_segments


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def single(layout, n, steps=1000):
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, n, 0)
    actions = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
    eng.rollout(actions, 100)
    ms = min(eng.rollout_timed(actions, steps) for _ in range(3))
    us = ms * 1e3 / steps
    gbs = algorithmic_bytes(S) * n / us / 1e3
    out = {"what": "one launch per step, outputs overwritten in place", "layout": layout, "kitchens": n,
           "us_per_launch": round(us, 2), "agent_env_steps_per_s": round(2 * n / us * 1e6), "algorithmic_GBps": round(gbs, 1),
           "roofline_frac": round(gbs / 8000, 3), "kernel_build": eng.kernel_build_info}
    eng.close()
    return out


def time_major(layout, n, T=200):
    """Observations/rewards/dones of every step written into their own time-major slot (what a trajectory
    buffer wants): the destination is T x 52 MB, far beyond the 256 MB Infinity Cache."""
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, n, 0)
    actions = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
    obs = torch.empty((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
    rew = torch.empty((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.empty((T, n), dtype=torch.int32, device="cuda")
    eng.rollout(actions, T, obs=obs, reward=rew, done=done)
    sec = timed(lambda: eng.rollout(actions, T, obs=obs, reward=rew, done=done))
    us = sec * 1e6 / T
    gbs = algorithmic_bytes(S) * n / us / 1e3
    out = {"what": "one launch per step, outputs into time-major trajectory slots", "layout": layout, "kitchens": n, "T": T,
           "trajectory_GB": round(obs.numel() / 1e9, 2), "us_per_launch": round(us, 2),
           "agent_env_steps_per_s": round(2 * n / us * 1e6), "algorithmic_GBps": round(gbs, 1), "roofline_frac": round(gbs / 8000, 3)}
    eng.close()
    return out


def config3(n=8192, k=7, T=400):
    """5 classic layouts x 8192 kitchens, 15 pairing segments each, engines stepping concurrently on 5 streams."""
    threads = n // (2 * k + 1)
    segs = pairing_segments(threads, k)
    engines, acts, streams = [], [], []
    for name in CLASSIC_LAYOUTS:
        engines.append(OvercookedEngine(get_base_layout_params(name, 200), n, 0))
        acts.append(torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32))
        streams.append(torch.cuda.Stream())

    def run():
        for eng, a, st in zip(engines, acts, streams):
            with torch.cuda.stream(st):
                eng.rollout(a, T)
    run()
    sec = timed(run)
    total_bytes = sum(algorithmic_bytes(e.size) * n for e in engines) * T
    out = {"what": "config 3: 5 classic layouts x 8192 kitchens, 15 pairing segments, 5 streams", "segments": len(segs),
           "kitchens_total": 5 * n, "us_per_joint_step": round(sec * 1e6 / T, 2),
           "agent_env_steps_per_s": round(2 * 5 * n * T / sec), "algorithmic_GBps": round(total_bytes / sec / 1e9, 1)}
    for e in engines:
        e.close()
    return out


if __name__ == "__main__":
    for r in (single("cramped_room", 65536), time_major("cramped_room", 65536), single("cramped_room", 4096),
              single("asymmetric_advantages", 8192), single("asymmetric_advantages", 65536),
              single("coordination_ring", 65536), single("counter_circuit", 65536), config3()):
        print(json.dumps(r), flush=True)

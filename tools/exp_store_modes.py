#!/usr/bin/env python3
"""Experiment: store mode of the plane stores (plain / nt / sc1 / sc0 sc1, see store16) x batch shape.
OC_STORE_MODE is read when a kernel is specialised, so every measurement builds fresh engines."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import algorithmic_bytes
from diverse_conventions_amd import CLASSIC_LAYOUTS, EngineGroup, OvercookedEngine, get_base_layout_params

os.environ["OC_JIT_CACHE"] = "0"


def single(layout, n, steps, time_major=False):
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, n, 0)
    actions = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
    if time_major:
        T = 100
        obs = torch.empty((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
        rew = torch.empty((T, 2, n), dtype=torch.int32, device="cuda")
        done = torch.empty((T, n), dtype=torch.int32, device="cuda")
        eng.rollout(actions, T, obs=obs, reward=rew, done=done)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.rollout(actions, T, obs=obs, reward=rew, done=done)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / T)
        us = best * 1e3
    else:
        eng.rollout(actions, 100)
        us = min(eng.rollout_timed(actions, steps) for _ in range(3)) * 1e3 / steps
    eng.close()
    return round(us, 2), round(algorithmic_bytes(S) * n / us / 1e3 / 8000, 3)


def group(n, steps):
    grp = EngineGroup.from_layouts(CLASSIC_LAYOUTS, n, 0)
    acts = [torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32) for _ in grp.engines]
    grp.rollout_timed(acts, 50)
    us = min(grp.rollout_timed(acts, steps) for _ in range(3)) * 1e3 / steps
    b = sum(algorithmic_bytes(e.size) * n for e in grp.engines)
    grp.close(close_engines=True)
    return round(us, 2), round(b / us / 1e3 / 8000, 3)


for mode in (sys.argv[1:] or ["0,1", "2,1", "2,2"]):
    os.environ["OC_STORE_MODE"] = mode
    out = {"mode(in_place,fresh)": mode,
           "cramped 65536 in place": single("cramped_room", 65536, 1000),
           "cramped 65536 time-major": single("cramped_room", 65536, 0, True),
           "asym 65536 in place": single("asymmetric_advantages", 65536, 400),
           "asym 8192": single("asymmetric_advantages", 8192, 1000),
           "cramped 4096": single("cramped_room", 4096, 1000),
           "config3 group 5x8192": group(8192, 400)}
    print(json.dumps(out), flush=True)

#!/usr/bin/env python3
"""EXPERIMENT: the policies-in-the-loop rollout with two launches per step against ONE (the env step folded into the
policy launch, OC_POLICY_FOLD=1), us per step:
    python tools/exp_fold_ab.py [layout] [kitchens]
Lines: fold 0 = two launches; fold 1 = one launch; debug 2 = one launch WITHOUT the env call (what the arrival bookkeeping costs;
results wrong by design); debug 4 = the env call made twice (is a call slow because its code is cold? no: twice the time)."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params
layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n_target = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
k = 7
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
P = (W - 2) * (H - 2)
threads = n_target // (2 * k + 1)
g2 = torch.Generator().manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g2) * 0.1
sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
        "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
        "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(k + 1)]
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
eng = OvercookedEngine(params, threads * (2 * k + 1), 0)
ro = FusedPairedRollout(eng, nets, threads, k)
for fold, dbg in (("0", None), ("1", None), ("1", "2"), ("1", "4")):
    os.environ["OC_POLICY_FOLD"] = fold
    if dbg: os.environ["OC_FOLD_DEBUG"] = dbg
    else: os.environ.pop("OC_FOLD_DEBUG", None)
    ro.step(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); ro.step(500); torch.cuda.synchronize()
    print(layout, eng.num_kitchens, "fold", fold, "debug", dbg, f"{(time.perf_counter()-t0)/500*1e6:.1f} us/step", flush=True)

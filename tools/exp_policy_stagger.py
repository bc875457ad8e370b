#!/usr/bin/env python3
"""Stagger of the two wavefronts that share a SIMD in the fused policy kernel (eight-wavefront workgroups: cramped_room).  Both run
the same program -- a matrix-dense position loop, then a vector- and latency-bound tail (fc2, head, epilogue) -- and start together:
delaying waves 4-7 puts one wave's tail beside its partner's loop (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
    OC_DIAGNOSTICS=1 OC_POLICY_DESYNC=<units of 64 cycles> python tools/exp_policy_stagger.py      (one process per value: read once)
Negative values ablate instead (timing only, outputs missing): -1 every block ends behind the head (no epilogue), -2 behind fc2, -3 behind
the position loop.  MI355X, 131 070 / 245 760 rows: 37.7 / 56.3 us; stagger 50..400: 37.3-40.1 / 56.2-58.4; ablations 35.8 / 33.2 / 31.4 and
53.8 / 50.9 / 46.6.  (A static `s_setprio 1` for waves 4-7, or for waves 0-3 -- the same guide's item 4 -- was tried the same way on
another box: 39.9 -> 42.6 / 39.9 us and 63.9 -> 63.5 / 64.0: nothing; not kept.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan

g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g) * 0.1
W, H, P = 5, 4, 6
sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P), "base.cnn.cnn.3.bias": r(64),
        "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64), ACTOR_HEAD + ".weight": r(6, 64), ACTOR_HEAD + ".bias": r(6)} for _ in range(8)]
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
out = {"desync": os.environ.get("OC_POLICY_DESYNC", "0")}
for threads in (4369, 8192):
    n = 2 * 15 * threads
    obs = (torch.rand((n, W, H, 20), device="cuda") < 0.1).to(torch.int8)
    segs = [(s * threads, threads, s % 8, TILE_ACT) for s in range(30)]
    plan = PolicyPlan(nets, n, segs, rows_per_tile=-torch.cuda.get_device_properties(0).multi_processor_count)
    acts = torch.zeros(n, dtype=torch.int32, device="cuda")
    lp = torch.zeros(n, device="cuda")
    for _ in range(20):
        plan.forward(obs, actions=acts, action_logp=lp, seed=1)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            plan.forward(obs, actions=acts, action_logp=lp, seed=1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    out[f"us_{n}_rows"] = round(best, 2)
print(json.dumps(out), flush=True)

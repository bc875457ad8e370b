#!/usr/bin/env python3
"""EXPERIMENT: the policies-in-the-loop rollout as TWO half-batches on two streams (each: policy launch -> env launch per step, all
steps enqueued from C), so that one half's env launch (memory bound) may run beside the other half's policy launch (matrix bound).
Against one batch on one stream, same total kitchens."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params
layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 65520
k = 7
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
P = (W - 2) * (H - 2)
g2 = torch.Generator().manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g2) * 0.1
sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
        "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
        "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(k + 1)]
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
G = 2 * k + 1
def make(n_parts):
    threads = total // n_parts // G
    out = []
    for _ in range(n_parts):
        eng = OvercookedEngine(params, threads * G, 0)
        out.append(FusedPairedRollout(eng, nets, threads, k))
    return out
steps = 400
for parts in (1, 2, 4):
    ros = make(parts)
    streams = [torch.cuda.Stream() for _ in ros]
    def run(n):
        for ro, st in zip(ros, streams):
            with torch.cuda.stream(st):
                ro.step(n)
    run(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    kitchens = sum(ro.engine.num_kitchens for ro in ros)
    print(f"{layout}: {parts} part(s) on {parts} stream(s), {kitchens} kitchens: {us:.1f} us per joint step = {2 * kitchens / us / 1e3:.2f} G agent-env steps/s", flush=True)
    for ro in ros:
        ro.engine.close()

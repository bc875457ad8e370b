"""Experiment: does splitting the batch in halves on two streams hide one half's front/tail under the other's stores?

    python tools/exp_two_halves.py [layout] [kitchens] [steps]

Whole batch as one engine (the headline configuration) against 2 and 4 engines of N/2, N/4 kitchens, each stepping on its
own stream (K launches enqueued from C per engine, all in flight together).  Reports wall-clock us per FULL step
(every kitchen advanced once).  A step of the real loop has a policy between env launches; this is the env-only bound.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from diverse_conventions_amd.engine import OvercookedEngine
from diverse_conventions_amd.layouts import get_base_layout_params


def run(layout, n, parts, steps):
    params = get_base_layout_params(layout, 400)
    m = n // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    engines, acts = [], []
    for s in streams:
        with torch.cuda.stream(s):
            engines.append(OvercookedEngine(params, m))
            acts.append(torch.randint(0, 6, (8, 2, m), dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()

    def go(k):
        for e, a, s in zip(engines, acts, streams):
            with torch.cuda.stream(s):
                e.rollout(a, k)
    go(50)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        go(steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    return best


def main():
    layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    for parts in (1, 2, 4):
        us = run(layout, n, parts, steps)
        print(f"{layout} {n} kitchens as {parts} engine(s) on {parts} stream(s): {us:.2f} us per full step "
              f"({2 * n / us * 1e6 / 1e9:.2f} G agent-env steps/s)", flush=True)


if __name__ == "__main__":
    main()

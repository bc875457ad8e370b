#!/usr/bin/env python3
"""Where the fixed cost of a 20-step timed run goes: the same 20 launches enqueued from C without events, with the two events
(oc_rollout_timed), and as a captured HIP graph; each bracketed by torch.cuda.synchronize() like bench.py's timed region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
n, K = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = OvercookedEngine(get_base_layout_params("cramped_room", 200), n, 0)
acts = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
eng.rollout(acts, 200); torch.cuda.synchronize()

def timed(fn, reps=7):
    out = []
    for _ in range(reps):
        eng.rollout(acts, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6)
    return sorted(out)

print("empty region (sync only):", [round(x, 1) for x in timed(lambda: None)])
print(f"{K} launches, no events:  ", [round(x, 1) for x in timed(lambda: eng.rollout(acts, K))])
print(f"{K} launches, two events: ", [round(x, 1) for x in timed(lambda: eng.rollout_timed(acts, K))])
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng.rollout(acts, K, stream=s) if "stream" in eng.rollout.__code__.co_varnames else None
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        eng.rollout(acts, K)
    print(f"{K} launches as a graph:  ", [round(x, 1) for x in timed(lambda: g.replay())])
except Exception as ex:
    print("graph capture failed:", ex)
# Measured (profiles/r03b_short_runs.log): no events 252 us, two marker events 266-273, the same events riding on the first / last
# dispatch packet (hipExtModuleLaunchKernel) 273, a 20-node graph 245; device-scope events (hipEventReleaseToDevice) change nothing.

#!/usr/bin/env python3
"""GAE op timing at the BASELINE trajectory size, against the torch formulation the reference uses."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd.rollout_ops import gae_returns

T, N, A = 200, 65536, 2
rewards = (torch.rand((T, N, A, 1), device="cuda") < 0.05).float() * 3
values = torch.randn((T + 1, N, A, 1), device="cuda")
masks = (torch.rand((T + 1, N, A, 1), device="cuda") > 0.005).float()
ret = torch.zeros_like(values)
for _ in range(3):
    gae_returns(rewards, values, masks, value_normalizer=(0.7, 2.5), returns=ret)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    gae_returns(rewards, values, masks, value_normalizer=(0.7, 2.5), returns=ret)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
M = N * A
bytes_alg = T * M * (4 + 4 + 4 + 4 + 4)        # r, v, mask read; returns, advantages written
# the reference's formulation: Python loop of torch ops (shared_buffer.py:286-300)
def torch_loop():
    gae = 0
    out = torch.zeros_like(values)
    dv = values * 2.5 + 0.7
    for s in reversed(range(T)):
        delta = rewards[s] + 0.99 * dv[s + 1] * masks[s + 1] - dv[s]
        gae = delta + 0.99 * 0.95 * masks[s + 1] * gae
        out[s] = gae + dv[s]
    return out
torch_loop(); torch.cuda.synchronize()
t0 = time.perf_counter(); ref = torch_loop(); torch.cuda.synchronize(); t_loop = (time.perf_counter() - t0) * 1e3
print(json.dumps({"op": "gae_returns", "T": T, "M": M, "ms": round(ms, 4), "algorithmic_GBps": round(bytes_alg / ms / 1e6, 1),
                  "roofline_frac": round(bytes_alg / ms / 1e6 / 8000, 3), "torch_python_loop_ms": round(t_loop, 2),
                  "bit_identical_to_torch_loop": bool(torch.equal(ref[:-1], ret[:-1]))}))

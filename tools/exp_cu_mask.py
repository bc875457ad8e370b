#!/usr/bin/env python3
"""Experiment: do CU-masked streams (hipExtStreamCreateWithCUMask) let several policy launches run side by side on disjoint
CU subsets of one MI355X?  Runs the same policy launch (a) alone on the whole device, (b) alone on a stream masked to half of
the CUs, (c) two copies concurrently on two complementary half-device streams."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_paired_rollout import random_actor
from diverse_conventions_amd import get_base_layout_params
from diverse_conventions_amd.fused_policy import TILE_ACT, PackedNetworks, PolicyPlan

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(cus):
    words = (ctypes.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


layout, n = "asymmetric_advantages", 8184
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_actor(W, H, gen=gen) for _ in range(4)])
rows = 2 * n
seg = rows // 4
obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)


def make(wgs):
    plan = PolicyPlan(nets, rows, [(i * seg, seg, i, TILE_ACT) for i in range(4)], rows_per_tile=-wgs)
    return plan, torch.zeros(rows, dtype=torch.int32, device="cuda"), torch.zeros(rows, device="cuda")


def timed(jobs, reps=30):
    """jobs: list of (stream, plan, acts, lp); launches all of them `reps` times, returns us per round."""
    for st, plan, a, lp in jobs:
        with torch.cuda.stream(st):
            plan.forward(obs, actions=a, action_logp=lp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for st, plan, a, lp in jobs:
            with torch.cuda.stream(st):
                plan.forward(obs, actions=a, action_logp=lp)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


full = torch.cuda.Stream()
lo = masked_stream([c for c in range(256) if (c // 4) % 2 == 0])      # alternate groups of 4 CUs
hi = masked_stream([c for c in range(256) if (c // 4) % 2 == 1])
lo2 = masked_stream(range(0, 128))
hi2 = masked_stream(range(128, 256))
p256, p128a, p128b = make(256), make(128), make(128)
out = {"alone, whole device (256 workgroups)": timed([(full,) + p256]),
       "alone, whole device, 128 workgroups": timed([(full,) + p128a]),
       "alone, half-device stream (interleaved mask), 128 workgroups": timed([(lo,) + p128a]),
       "two at once, complementary interleaved masks, 128 workgroups each": timed([(lo,) + p128a, (hi,) + p128b]),
       "two at once, masks 0-127 / 128-255": timed([(lo2,) + p128a, (hi2,) + p128b]),
       "two at once, two unmasked streams, 256 workgroups each": timed([(full,) + p256, (torch.cuda.Stream(),) + make(256)]),
       "two one after the other, one stream": timed([(full,) + p256, (full,) + make(256)])}
print(json.dumps(out, indent=1))

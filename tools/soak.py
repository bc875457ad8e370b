#!/usr/bin/env python3
"""Soak: long random rollouts of the step kernels against the C oracle, full state and last observation compared every chunk --
in place, into rotating time-major slots (the halo builds at odd kitchen counts read one record block and write the other), K steps
per launch, and the N-player kernel.  A race between wavefronts or launches that the short parity tests never provoke would show as
a state mismatch here.    python tools/soak.py [chunks=12] [chunk_steps=400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from oracle import oracle

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
CASES = [("cramped_room", None, 65536, "in_place"), ("cramped_room", None, 65535, "time_major"), ("asymmetric_advantages", None, 49157, "time_major"),
         ("coordination_ring", None, 65536, "time_major"), ("cramped_room", None, 65536, "fused"), ("multiplayer_schelling", None, 36001, "in_place"),
         ("multiplayer_schelling", 3, 33001, "time_major"), ("simple_single", None, 70003, "fused")]
g = torch.Generator(device="cuda").manual_seed(123)
bad = 0
for layout, maxp, n, mode in CASES:
    params = get_base_layout_params(layout, 200, max_num_players=maxp)
    P = params["num_players"]
    eng = OvercookedEngine(params, n, 0)
    ora = oracle.OracleVecEnv(params, n, nthreads=16)
    slots = 5
    tm = None
    if mode == "time_major":
        tm = (torch.zeros((slots,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda"), torch.zeros((slots, P, n), dtype=torch.int32, device="cuda"),
              torch.zeros((slots, n), dtype=torch.int32, device="cuda"))
    t0 = time.time()
    ok = True
    for c in range(chunks):
        acts = torch.randint(0, 6, (steps, P, n), device="cuda", dtype=torch.int32, generator=g)
        acts[:, :, ::4] = torch.where(torch.rand((steps, P, (n + 3) // 4), device="cuda", generator=g) < 0.4, 5, acts[:, :, ::4])   # busier kitchens
        if mode == "in_place":
            eng.rollout(acts, steps)
            last = eng.obs
        elif mode == "fused":
            eng.rollout_fused(steps, actions=acts)
            last = eng.obs
        else:
            for s0 in range(0, steps, slots):                       # rotating slots: every launch a fresh destination
                k = min(slots, steps - s0)
                eng.rollout(acts[s0:s0 + k], k, obs=tm[0][:k], reward=tm[1][:k], done=tm[2][:k])
            last = tm[0][(steps - 1) % slots if steps % slots else slots - 1]
        host = acts.cpu().numpy()
        for t in range(steps):
            oobs, _, _ = ora.step(host[t])
        same = all(np.array_equal(x, y) for x, y in zip(eng.get_state(), ora.get_state())) and np.array_equal(last.cpu().numpy(), oobs)
        ok = ok and same
        if not same:
            print(f"MISMATCH {layout} P={P} n={n} {mode} chunk {c}", flush=True)
            bad += 1
            break
    print(f"{layout} P={P} n={n} {mode}: {chunks * steps} steps {'bit-exact' if ok else 'FAILED'} ({time.time() - t0:.0f} s; {eng.kernel_build_info[:60]})", flush=True)
    eng.close()
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""bench.py's config4_share_policy_in_loop leg alone (asymmetric_advantages, 8184 kitchens, 16 conventions in 31 pairing segments, actors
and one critic per segment on both sides in ONE policy launch per step + the env launch), for A/B runs of kernel builds and of the
OC_POLICY_MONO / OC_POLICY_COOP switches (read per launch):   python tools/exp_config4_share.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sk, sthreads = 15, (65536 // 8) // 31
sG = 2 * sk + 1
sp = get_base_layout_params("asymmetric_advantages", 200)
sW, sH = sp["width"], sp["height"]
sP = (sW - 2) * (sH - 2)
g3 = torch.Generator().manual_seed(1)
r3 = lambda *sh: torch.randn(*sh, generator=g3) * 0.1
base3 = lambda: {"base.cnn.cnn.0.weight": r3(32, 20, 3, 3), "base.cnn.cnn.0.bias": r3(32), "base.cnn.cnn.3.weight": r3(64, 32 * sP),
                 "base.cnn.cnn.3.bias": r3(64), "base.cnn.cnn.5.weight": r3(64, 64), "base.cnn.cnn.5.bias": r3(64)}
actors3 = [dict(base3(), **{"act.action_out.linear.weight": r3(6, 64), "act.action_out.linear.bias": r3(6)}) for _ in range(sk + 1)]
critics3 = [dict(base3(), **{"v_out.weight": r3(1, 64), "v_out.bias": r3(1)}) for _ in range(sG)]
nets3 = PackedNetworks.from_state_dicts((sW, sH, 20), "cuda", actors=actors3, critics=critics3)
eng3 = OvercookedEngine(sp, sthreads * sG, 0)
ro3 = FusedPairedRollout(eng3, nets3, sthreads, sk, critic_nets=[[sk + 1 + g for g in range(sG)]] * 2)
ro3.step(200)
torch.cuda.synchronize()
for mode in ({}, {"OC_POLICY_MONO": "0"}, {"OC_POLICY_COOP": "0"}, {"OC_POLICY_MONO": "0", "OC_POLICY_COOP": "0"}, {}):
    for k in ("OC_POLICY_MONO", "OC_POLICY_COOP"):
        os.environ.pop(k, None)
    os.environ.update(mode)
    best = float("inf")
    for _ in range(3):
        t3 = time.perf_counter()
        ro3.step(steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t3) / steps * 1e6)
    print(f"{str(mode) if mode else 'default':60s} {best:7.2f} us per step", flush=True)

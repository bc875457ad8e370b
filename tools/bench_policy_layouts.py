#!/usr/bin/env python3
"""Policy kernel (oc_policy_forward) on every classic layout: the launch alone (4 actors + 4 critics, both sides) and the
whole policies-in-the-loop step of config 4's per-GPU share.  The bound printed is the MFMA issue time of the launch:
per 32 rows P x (6 x digits + 12) + 24 + 12 matrix instructions of 32 cycles each on one SIMD (i8 conv digits + f16
two-piece products of the dense layers), at 2.4 GHz and at the ~2.1 GHz the chip holds under this load.
Round-2 history: the same script compared the removed two-wavefront plans (profiles/r02_policy_pair_split.log,
r02_policy_conv_dense_split.log; f32 dense layers then, 23 040 cycles per 32 rows on cramped_room against 11 520 now)."""
import json, os, sys, time
if os.environ.get("OC_POLICY_PLAN"):
    os.environ.setdefault("OC_DIAGNOSTICS", "1")      # the plan override is a knob of the diagnostics build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_paired_rollout import random_actor, random_critic, run_fused
from diverse_conventions_amd import get_base_layout_params
from diverse_conventions_amd.fused_policy import TILE_ACT, PackedNetworks, PolicyPlan

plan = os.environ.get("OC_POLICY_PLAN", "default")
DIGITS = 3                                                # the default: three base-256 conv weight digits
for layout, n in (("cramped_room", 61440), ("asymmetric_advantages", 30720), ("counter_circuit", 30720), ("forced_coordination", 61440), ("coordination_ring", 61440)):
    params = get_base_layout_params(layout, 200)
    W, H = params["width"], params["height"]
    P = (W - 2) * (H - 2)
    gen = torch.Generator().manual_seed(0)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_actor(W, H, gen=gen) for _ in range(4)],
                                           critics=[random_critic(W, H, gen) for _ in range(4)])
    rows = 2 * n
    seg = rows // 4
    segs = [(i * seg, seg, i, TILE_ACT) for i in range(4)] + [(i * seg, seg, 4 + i, 0) for i in range(4)]
    pp = PolicyPlan(nets, rows, segs)
    obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
    acts = torch.zeros(rows, dtype=torch.int32, device="cuda")
    lp = torch.zeros(rows, device="cuda")
    val = torch.zeros(rows, device="cuda")
    for _ in range(5):
        pp.forward(obs, raw=val, actions=acts, action_logp=lp)
    torch.cuda.synchronize()
    us = 1e9
    for _ in range(5):                                    # best of five: run-to-run clock differences are a few per cent
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            pp.forward(obs, raw=val, actions=acts, action_logp=lp)
        e1.record()
        torch.cuda.synchronize()
        us = min(us, e0.elapsed_time(e1) * 1e3 / 30)
    evals = 2 * rows
    cycles = evals / 32 * 32 * (P * (6 * DIGITS + 12) + 24 + 12)
    bound24 = cycles / 1024 / 2400.0
    print(json.dumps({"plan": plan, "layout": layout, "evaluations": evals, "us": round(us, 1), "issue_bound_us_at_2.4GHz": round(bound24, 1),
                      "frac_of_bound_2.4GHz": round(bound24 / us, 3), "frac_at_2.1GHz": round(bound24 * 2.4 / 2.1 / us, 3)}), flush=True)
# BASELINE config 4 with the policies in the loop (asymmetric_advantages, 16 conventions in 31 pairing segments, critics): one GPU's share
# of the 65 536 kitchens on 8, 4, 2 GPUs and the whole batch on one -- the strong-scaling prediction of DESIGN.md section 7
for gpus in ((8, 4, 2, 1) if not os.environ.get("OC_SKIP_SHARES") else ()):
    r = run_fused("asymmetric_advantages", (65536 // gpus) // 31, 15, critics=True)
    print(json.dumps({"plan": plan, "config4_share": r, "of_gpus": gpus}), flush=True)

import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_paired_rollout import random_actor, random_critic
from diverse_conventions_amd import get_base_layout_params
from diverse_conventions_amd.fused_policy import TILE_ACT, PackedNetworks, PolicyPlan
layout, n = "asymmetric_advantages", 30720
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_actor(W, H, gen=gen) for _ in range(4)], critics=[random_critic(W, H, gen) for _ in range(4)])
rows = 2 * n; seg = rows // 4
segs = [(i * seg, seg, i, TILE_ACT) for i in range(4)] + [(i * seg, seg, 4 + i, 0) for i in range(4)]
pp = PolicyPlan(nets, rows, segs)
obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
acts = torch.zeros(rows, dtype=torch.int32, device="cuda"); lp = torch.zeros(rows, device="cuda"); val = torch.zeros(rows, device="cuda")
for _ in range(5): pp.forward(obs, raw=val, actions=acts, action_logp=lp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): pp.forward(obs, raw=val, actions=acts, action_logp=lp)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("OC_POLICY_PLAN"), "desync", os.environ.get("OC_POLICY_DESYNC"), round(e0.elapsed_time(e1) * 1e3 / 30, 1), "us")

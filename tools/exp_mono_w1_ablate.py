import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
from bench_fused_policy import random_net
W,H=9,5
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, gen) for _ in range(8)])
rows=131072
obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
plan = PolicyPlan(nets, rows, [(g * 4096, 4096, g % 8, TILE_ACT) for g in range(32)])
actions = torch.zeros(rows, dtype=torch.int32, device="cuda")
def timed():
    for _ in range(10): plan.forward(obs, actions=actions, deterministic=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): plan.forward(obs, actions=actions, deterministic=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*10
for m in ("0","1","2","1","2","0"):
    os.environ["OC_POLICY_MONO"]=m
    print("OC_POLICY_MONO", m, round(timed(),2), "us")

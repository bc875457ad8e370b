#!/usr/bin/env python3
"""Config 3's shape with the policies in the loop, COLLECTING into time-major slots (what training does) against stepping in place:
five layouts x 8190 kitchens, 8 conventions with critics, ONE policy launch over the five grids + ONE env launch per joint step.
OC_PLAIN_FRESH=0 keeps the streaming stores for the fresh slots."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from diverse_conventions_amd import CLASSIC_LAYOUTS, EngineGroup, GroupPairedRollout, TrajectoryBuffer
from exp_group_streams import make_nets
k, n, T = 7, 8190, 100
threads = n // (2 * k + 1)
gen = torch.Generator().manual_seed(0)
nets, cn = make_nets(CLASSIC_LAYOUTS, k, True, gen)
grp = EngineGroup.from_layouts(CLASSIC_LAYOUTS, threads * (2 * k + 1), 0)
ro = GroupPairedRollout(grp, nets, threads, k, critic_nets=cn, seed=5)
ro.step(20); torch.cuda.synchronize()
t0 = time.perf_counter(); ro.step(300); torch.cuda.synchronize()
in_place = (time.perf_counter() - t0) / 300 * 1e6
trajs = [TrajectoryBuffer(e, T) for e in grp.engines]
ro.collect(trajs); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); ro.collect(trajs); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / T * 1e6)
print(json.dumps({"plain_fresh": os.environ.get("OC_PLAIN_FRESH", "1"), "in_place_us_per_joint_step": round(in_place, 1), "collect_time_major_us_per_joint_step": round(best, 1)}))

#!/usr/bin/env python3
"""Interleaved A/B of launch geometries for the headline workload (cramped_room, 65 536 kitchens, in place): each candidate is
re-measured `rounds` times in rotation so that drift between engines / clocks averages out.  Prints median and min us/launch."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cands = [tuple(int(v) for v in c.split(",")) for c in (sys.argv[3:] or ["32,4,8", "32,2,8", "16,2,12", "16,3,8", "16,2,8"])]
params = get_base_layout_params(layout, 200)
acts = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
engines = []
for cand in cands:
    kpw, gp, wpc = cand[:3]
    if len(cand) > 3:                                     # 4th field: OC_FAST_FLUSH for this engine's specialised build
        os.environ["OC_FAST_FLUSH"] = str(cand[3])
    else:
        os.environ.pop("OC_FAST_FLUSH", None)
    if len(cand) > 4:                                     # 5th field: passes of the first flush group (0: as the others)
        os.environ["OC_FIRST_PASSES"] = str(cand[4])
    else:
        os.environ.pop("OC_FIRST_PASSES", None)
    if len(cand) > 5:                                     # 6th field: wavefronts per workgroup (1, 2, 4, 8)
        os.environ["OC_WG_WAVES"] = str(cand[5])
    else:
        os.environ.pop("OC_WG_WAVES", None)
    if len(cand) > 6:                                     # 7th field: OC_ENCODE_ILP4 (experiment)
        os.environ["OC_ENCODE_ILP4"] = str(cand[6])
    else:
        os.environ.pop("OC_ENCODE_ILP4", None)
    e = OvercookedEngine(params, n, 0)
    e.set_tuning(kpw, gp, wpc)
    e.rollout(acts, 100)
    engines.append(e)
res = [[] for _ in cands]
for r in range(9):
    for i, e in enumerate(engines):
        res[i].append(e.rollout_timed(acts, 1000) * 1e3 / 1000)
for c, r in zip(cands, res):
    print(f"{layout} N={n} kpw,passes,wpc[,fast flush]={c}: median {statistics.median(r):.3f} us  min {min(r):.3f}  max {max(r):.3f}")

#!/usr/bin/env python3
"""Launch geometries for the time-major destination (every step's planes into their own slot, non-temporal stores):
interleaved A/B like tools/ab_geometry.py.  Candidates kpw,passes,wpc[,fast flush[,fresh store mode]]."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cands = [tuple(int(v) for v in c.split(",")) for c in (sys.argv[3:] or ["16,2,12", "32,2,12"])]
params = get_base_layout_params(layout, 200)
slots = 64
acts = torch.randint(0, 6, (slots, 2, n), device="cuda", dtype=torch.int32)
engines = []
for cand in cands:
    if len(cand) > 3:
        os.environ["OC_FAST_FLUSH"] = str(cand[3])
    else:
        os.environ.pop("OC_FAST_FLUSH", None)
    if len(cand) > 4:                                     # fifth field: store mode of fresh destinations (0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt)
        os.environ["OC_STORE_MODE"] = f"2,{cand[4]}"
    else:
        os.environ.pop("OC_STORE_MODE", None)
    if len(cand) > 5:                                     # sixth field: passes of the FIRST flush group (0: as the others)
        os.environ["OC_FIRST_PASSES"] = str(cand[5])
    else:
        os.environ.pop("OC_FIRST_PASSES", None)
    if len(cand) > 7:                                     # seventh, eighth field: kitchens per group on even / odd XCDs (0,0: equal groups)
        os.environ["OC_XCC_WEIGHTS"] = ",".join(str(cand[6 + (x & 1)]) for x in range(8)) if cand[6] else ""
    elif "OC_XCC_WEIGHTS" in os.environ and not os.environ.get("OC_XCC_WEIGHTS_KEEP"):
        os.environ.pop("OC_XCC_WEIGHTS")
    e = OvercookedEngine(params, n, 0)
    if cand[0]:                                           # (0,0,0: the engine's own choice, fresh geometry included)
        e.set_tuning(*cand[:3])
    engines.append(e)
W, H = params["width"], params["height"]
obs = torch.zeros((slots, 2, n, W, H, 20), dtype=torch.int8, device="cuda")
rew = torch.zeros((slots, 2, n), dtype=torch.int32, device="cuda")
done = torch.zeros((slots, n), dtype=torch.int32, device="cuda")
res = [[] for _ in cands]
for r in range(7):
    for i, e in enumerate(engines):
        kw = dict(obs=obs) if os.environ.get("OC_AB_OBS_ONLY") else dict(obs=obs, reward=rew, done=done)   # (obs only: rewards / dones in place)
        e.rollout(acts, slots, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            e.rollout(acts, slots, **kw)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / (8 * slots) * 1e6)
for c, r in zip(cands, res):
    print(f"{layout} N={n} time-major kpw,passes,wpc[,fast flush]={c}: median {statistics.median(r):.3f} us  min {min(r):.3f}")

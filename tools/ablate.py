#!/usr/bin/env python3
"""Phase-cost diagnostics for the fused step kernel (timing only, outputs are wrong when a phase is
skipped): OC_ABLATE bits 1=skip transition 2=skip obs stores 4=skip encode compute 8=skip state
write-back 16=skip state load."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
kpw = int(sys.argv[3]) if len(sys.argv) > 3 else 16
g = int(sys.argv[4]) if len(sys.argv) > 4 else 2
w = int(sys.argv[5]) if len(sys.argv) > 5 else 8
params = get_base_layout_params(layout, 200)
actions = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
fresh = len(sys.argv) > 6 and sys.argv[6] == "fresh"       # step into time-major slots instead of in place
W, H = params["width"], params["height"]
slots = (torch.zeros((64, 2, n, W, H, 20), dtype=torch.int8, device="cuda"), torch.zeros((64, 2, n), dtype=torch.int32, device="cuda"),
         torch.zeros((64, n), dtype=torch.int32, device="cuda")) if fresh else None
for ab in (0, 1, 2, 4, 8, 16, 6, 7, 1 | 4, 2 | 8, 31, 31 - 2, 31 - 4):
    os.environ["OC_ABLATE"] = str(ab)
    eng = OvercookedEngine(params, n, 0)
    if kpw:
        eng.set_tuning(kpw, g, w)
    if fresh:
        import time
        eng.rollout(actions, 64, obs=slots[0], reward=slots[1], done=slots[2]); torch.cuda.synchronize()
        us = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5): eng.rollout(actions, 64, obs=slots[0], reward=slots[1], done=slots[2])
            torch.cuda.synchronize(); us = min(us, (time.perf_counter() - t0) / 320 * 1e6)
    else:
        eng.rollout(actions, 50)
        us = min(eng.rollout_timed(actions, 300) for _ in range(3)) * 1e3 / 300
    print(f"ablate={ab:2d} ({ab:05b})  {us:7.2f} us/launch", flush=True)
    eng.close()

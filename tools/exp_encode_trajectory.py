#!/usr/bin/env python3
"""oc_encode_trajectory (the receive side of the exchange) at the per-GPU kitchen counts of config 4: one launch per peer for T = 200 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
params = get_base_layout_params("asymmetric_advantages", 200)
T = 200
for n in (65536, 16384, 8192):
    eng = OvercookedEngine(params, n, 0)
    rec = eng.export_states().unsqueeze(0).repeat(T, 1, 1).contiguous()
    out = torch.empty((T, 2, n, 9, 5, 20), dtype=torch.int8, device="cuda")
    for _ in range(2):
        eng.encode_trajectory_into(rec, out, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.encode_trajectory_into(rec, out, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"n={n}: {ms:.3f} ms per peer-trajectory, {out.numel() / ms / 1e6:.0f} GB/s written ({eng.kernel_build_info})", flush=True)
    eng.close()

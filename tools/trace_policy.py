#!/usr/bin/env python3
"""Per-wavefront phase timeline of ONE oc_policy_forward launch (s_memrealtime stamps, 100 MHz, of each wave's
FIRST 32-row block; plus the shader clock over the wave's whole life).  Diagnostic only."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diverse_conventions_amd import native
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_fused_policy import random_net

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 4)
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 245760
rpt = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
digits = int(sys.argv[5]) if len(sys.argv) > 5 else 3
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, gen) for _ in range(4)], digits=digits)
obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
seg = rows // 4
plan = PolicyPlan(nets, rows, [(g * seg, seg, g, TILE_ACT) for g in range(4)], rows_per_tile=rpt)
actions = torch.zeros(rows, dtype=torch.int32, device="cuda")
for _ in range(3):
    plan.forward(obs, actions=actions)
torch.cuda.synchronize()
L = native.lib()
L.oc_policy_debug_trace.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
assert L.oc_policy_debug_trace(1, None, 0) == 0
plan.forward(obs, actions=actions)
nw = 256 * 8
buf = np.zeros((nw, 32), dtype=np.uint64)
assert L.oc_policy_debug_trace(0, buf.ctypes.data_as(ctypes.c_void_p), nw) == 0
buf = buf[buf[:, 0] > 0]
t = buf[:, :9].astype(np.float64) / 100.0
t0 = t[:, 0].min()
names = ["start", "weights+regs in (barrier passed)", "block 0: before row wait", "rows landed", "conv+fc1 done", "fc2 done",
         "head done", "epilogue done (block 0)", "wave end"]
print(f"{W}x{H} rows={rows} rows_per_tile={rpt} digits={digits}: {plan.n_tiles} tiles, {len(buf)} waves")
for i, nm in enumerate(names):
    c = t[:, i] - t0
    print(f"  {nm:34s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
pro = buf[:, [9, 10, 11, 12, 13, 1]].astype(np.float64) / 100.0
if (pro[:, 0] > 0).all():
    pn = ["kernel entry", "stretch found, first tile read", "first rows' DMA issued", "weight DMA issued", "weight DMA landed", "barrier passed"]
    print("  prologue (median us after the launch's first stamp; step):")
    prev = None
    for i, nm in enumerate(pn):
        c = pro[:, i] - pro[:, 0].min()
        print(f"    {nm:34s} {np.median(c):6.2f}" + (f"   (+{np.median(pro[:, i] - pro[:, i - 1]):.2f})" if i else ""))
d = t[:, 1:9] - t[:, 0:8]
print("  phase durations (median us):", " ".join(f"{np.median(d[:, i]):.2f}" for i in range(8)))
fine = buf[buf[:, 30] > 0][:, 16:31].astype(np.float64) / 100.0
if len(fine):
    names2 = ["rows wait", "frags(0)+conv(0) issue+frags(1)", "combine(0)"] + [f"position {i}: conv(p+1)+fc1(p)" for i in range(6)] + \
             ["(loop exit)", "dense operands + next rows DMA issue", "fc2", "head", "epilogue"]
    d2 = fine[:, 1:] - fine[:, :-1]
    print(f"  second block of {len(fine)} waves, median us per phase (two waves share each SIMD):")
    for nm, v in zip(names2, np.median(d2, axis=0)):
        print(f"    {nm:42s} {v:6.2f}")
    print(f"    {'whole block':42s} {np.median(fine[:, -1] - fine[:, 0]):6.2f}")
    extra = buf[buf[:, 30] > 0][:, 31].astype(np.float64) / 100.0
    if extra.any():
        print(f"    {'  of which: bias reads (before the DMA issue)':42s} {np.median(extra - fine[:, 10]):6.2f}")
life_us = t[:, 8] - t[:, 0]
cyc = (buf[:, 15] - buf[:, 14]).astype(np.float64)
print(f"  shader clock over wave life: median {np.median(cyc / life_us):.0f} MHz")

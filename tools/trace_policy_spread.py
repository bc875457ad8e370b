#!/usr/bin/env python3
"""Where the spread between a policy launch's wavefronts comes from (diagnostics build: the per-wavefront phase stamps of
tools/trace_policy.py, looked at per workgroup, per XCD and per SIMD partner instead of as medians).  The launch ends with its LAST
wavefront: the bench line's launch (15 segments x 4369 rows x 2 sides, 8 networks) by default.  Diagnostic only."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diverse_conventions_amd import native
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, TILE_ACT, PackedNetworks, PolicyPlan
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_fused_policy import random_net

W, H = 5, 4
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 4369
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = seg * nseg
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, gen) for _ in range(8)])
obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
plan = PolicyPlan(nets, rows, [(g * seg, seg, g % 8, TILE_ACT) for g in range(nseg)])
actions = torch.zeros(rows, dtype=torch.int32, device="cuda")
for _ in range(5):
    plan.forward(obs, actions=actions)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    plan.forward(obs, actions=actions)
e1.record()
torch.cuda.synchronize()
print(f"{rows} rows, {plan.n_tiles} tiles: {e0.elapsed_time(e1) / 50 * 1000:.1f} us per launch back to back")
L = native.lib()
L.oc_policy_debug_trace.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
assert L.oc_policy_debug_trace(1, None, 0) == 0
plan.forward(obs, actions=actions)
nw = 256 * 8
buf = np.zeros((nw, 32), dtype=np.uint64)
assert L.oc_policy_debug_trace(0, buf.ctypes.data_as(ctypes.c_void_p), nw) == 0
ids = np.arange(nw)
ok = buf[:, 0] > 0
buf, ids = buf[ok], ids[ok]
wg, wave = ids // 8, ids % 8
t = buf[:, :9].astype(np.float64) / 100.0
t -= t[:, 0].min()
start, ready, b0done, end = t[:, 0], t[:, 1], t[:, 7], t[:, 8]
print(f"waves {len(ids)}; start med {np.median(start):.2f} max {start.max():.2f}; ready med {np.median(ready):.2f} max {ready.max():.2f}; "
      f"block-0 done med {np.median(b0done):.2f} max {b0done.max():.2f}; end med {np.median(end):.2f} max {end.max():.2f}")
# per workgroup
wgs = np.unique(wg)
wg_end = np.array([end[wg == g].max() for g in wgs])
wg_first = np.array([end[wg == g].min() for g in wgs])
wg_start = np.array([start[wg == g].min() for g in wgs])
print(f"per workgroup: last wave's end med {np.median(wg_end):.2f} max {wg_end.max():.2f} min {wg_end.min():.2f}; "
      f"spread inside a workgroup (last - first end) med {np.median(wg_end - wg_first):.2f} max {(wg_end - wg_first).max():.2f}")
print(f"  workgroup start med {np.median(wg_start):.2f} max {wg_start.max():.2f}; corr(start, end) {np.corrcoef(wg_start, wg_end)[0, 1]:.2f}")
print("  per XCD (workgroup % 8): " + "  ".join(f"{x}: end med {np.median(wg_end[wgs % 8 == x]):.1f} max {wg_end[wgs % 8 == x].max():.1f}" for x in range(8)))
life = end - start
print(f"wave life med {np.median(life):.2f} min {life.min():.2f} max {life.max():.2f}")
for w in range(8):
    m = wave == w
    if m.any():
        print(f"  wave {w}: n {m.sum()}  ready {np.median(ready[m]):.2f}  block-0 done {np.median(b0done[m]):.2f}  end med {np.median(end[m]):.2f} max {end[m].max():.2f}")
cyc = (buf[:, 15] - buf[:, 14]).astype(np.float64)
mhz = cyc / life
print(f"shader clock over wave life: med {np.median(mhz):.0f} min {mhz.min():.0f} max {mhz.max():.0f} MHz; per XCD "
      + " ".join(f"{np.median(mhz[wg % 8 == x]):.0f}" for x in range(8)))
worst = np.argsort(-wg_end)[:8]
print("slowest workgroups: " + ", ".join(f"wg {wgs[i]} (xcd {wgs[i] % 8}) {wg_end[i]:.1f}" for i in worst))

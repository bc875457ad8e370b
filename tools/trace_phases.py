#!/usr/bin/env python3
"""Per-wave phase timeline of ONE step launch (diagnostic build path: s_memrealtime stamps, 100 MHz).
Stamps: 0 wave start, 1 prologue done (tables + first state block landed), 2 first transition done,
3 first plane flush issued, 4 last plane flush issued, 5 wave end.  Times in us relative to the
earliest wave start."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params, native

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
kpw = int(sys.argv[3]) if len(sys.argv) > 3 else 32
g = int(sys.argv[4]) if len(sys.argv) > 4 else 2
w = int(sys.argv[5]) if len(sys.argv) > 5 else 8
fresh = len(sys.argv) > 6 and sys.argv[6] == "fresh"      # trace a step into a time-major slot (destination != the previous step's)
params = get_base_layout_params(layout, 200)
eng = OvercookedEngine(params, n, 0)
eng.set_tuning(kpw, g, w)
actions = torch.randint(0, 6, (8, 2, n), device="cuda", dtype=torch.int32)
slots = None
if fresh:
    W, H = params["width"], params["height"]
    slots = (torch.zeros((64, 2, n, W, H, 20), dtype=torch.int8, device="cuda"), torch.zeros((64, 2, n), dtype=torch.int32, device="cuda"),
             torch.zeros((64, n), dtype=torch.int32, device="cuda"))
    eng.rollout(actions, 64, obs=slots[0], reward=slots[1], done=slots[2])
    if kpw == 0:
        kpw, g, w = 16, 4, 8                           # the engine's own geometry for fresh destinations at this size (set_tuning(0,0,0) keeps it)
eng.rollout(actions, 20) if not fresh else None
torch.cuda.synchronize()
L = native.lib()
L.oc_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
native.check(L.oc_debug_trace(eng._handle, 1, None, 0))
if fresh:
    eng.rollout(actions, 7, obs=slots[0][:7], reward=slots[1][:7], done=slots[2][:7])     # the traced launch follows launches of its own kind
    eng.step(actions[0], obs=slots[0][7], reward=slots[1][7], done=slots[2][7])
else:
    eng.step(actions[0])
nw = min((n + kpw - 1) // kpw, w * 256)
buf = np.zeros((nw, 16), dtype=np.uint64)
native.check(L.oc_debug_trace(eng._handle, 0, buf.ctypes.data_as(ctypes.c_void_p), nw))
order = [0, 1, 6, 7, 8, 2, 3, 4, 5]
t = buf[:, order].astype(np.float64) / 100.0          # 100 MHz -> us
t0 = t[:, 0].min()
names = ["start", "prologue", "T: lds fill", "T: loads", "T: interacts", "transition", "first flush", "last flush", "end"]
print(f"{layout} N={n} kpw={kpw} passes={g} waves/CU={w}: {nw} waves")
for i, nm in enumerate(names):
    c = t[:, i] - t0
    print(f"  {nm:12s} min {c.min():6.2f}  p10 {np.percentile(c,10):6.2f}  median {np.median(c):6.2f}  p90 {np.percentile(c,90):6.2f}  max {c.max():6.2f}")
d = np.diff(t, axis=1)
for i in range(len(names) - 1):
    print(f"  {names[i]:>12s} -> {names[i+1]:12s} median {np.median(d[:, i]):6.2f} us   p90 {np.percentile(d[:, i],90):6.2f}")
xcc = buf[:, 10]
print("  waves per XCC id:", np.bincount(xcc.astype(np.int64) & 0xF))
eng.close()

# where do the slow waves live?  end time by XCD and by dispatch order
end = t[:, -1] - t0
for x in range(8):
    sel = (xcc.astype(np.int64) & 0xF) == x
    print(f"  XCC {x}: start median {np.median(t[sel, 0] - t0):5.2f}  end median {np.median(end[sel]):5.2f}  max {end[sel].max():5.2f}")
q = np.array_split(np.arange(nw), 8)
print("  end median by wave-id octile:", [round(float(np.median(end[i])), 2) for i in q])
print("  start median by wave-id octile:", [round(float(np.median((t[:, 0] - t0)[i])), 2) for i in q])
half = nw // 2
for nm, sel in (("first-dispatched half ", slice(0, half)), ("second-dispatched half", slice(half, nw))):
    print(f"  {nm}: " + "  ".join(f"{names[i]} {np.median(t[sel, i] - t0):.2f}" for i in range(len(names))))

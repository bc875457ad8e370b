#!/usr/bin/env python3
"""EXPERIMENT: phase stamps (s_memrealtime, 100 MHz) of the env step body when it runs folded into the policy launch
(OC_POLICY_FOLD=1, OC_FOLD_TRACE=1): one sample per workgroup (the last env call of the launch), phases as tools/trace_phases.py."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import ctypes, os, sys
os.environ["OC_FOLD_TRACE"] = "1"
os.environ["OC_POLICY_FOLD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params, native
layout = sys.argv[1] if len(sys.argv) > 1 else "asymmetric_advantages"
n_target = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = 7
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
P = (W - 2) * (H - 2)
threads = n_target // (2 * k + 1)
g2 = torch.Generator().manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g2) * 0.1
sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
        "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
        "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(k + 1)]
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
eng = OvercookedEngine(params, threads * (2 * k + 1), 0)
L = native.lib()
native.check(L.oc_debug_trace(eng._handle, 1, None, 0))          # before the first folded rollout: the template then carries the buffer
ro = FusedPairedRollout(eng, nets, threads, k)
ro.step(3)
torch.cuda.synchronize()
nw = 256
buf = np.zeros((nw, 16), dtype=np.uint64)
native.check(L.oc_debug_trace(eng._handle, 0, buf.ctypes.data_as(ctypes.c_void_p), nw))
order = [0, 1, 6, 7, 8, 2, 3, 4, 5]
names = ["start", "prologue", "T: lds fill", "T: loads", "T: interacts", "transition", "first flush", "last flush", "end"]
t = buf[:, order].astype(np.float64) / 100.0
ok = (buf[:, 0] > 0) & (buf[:, 5] > 0)
t = t[ok]
print(f"{layout} {eng.num_kitchens} kitchens: {ok.sum()} workgroups with a sample")
d = np.diff(t, axis=1)
for i in range(len(names) - 1):
    print(f"  {names[i]:>12s} -> {names[i+1]:12s} median {np.median(d[:, i]):7.2f} us   p90 {np.percentile(d[:, i], 90):7.2f}")
print(f"  whole call: median {np.median(t[:, -1] - t[:, 0]):.2f} us")

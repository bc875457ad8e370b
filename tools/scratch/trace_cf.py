import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench_paired_rollout import random_actor, random_critic
from diverse_conventions_amd import get_base_layout_params, native
from diverse_conventions_amd.fused_policy import TILE_ACT, PackedNetworks, PolicyPlan
layout, n = "asymmetric_advantages", 30720
params = get_base_layout_params(layout, 200)
W, H = params["width"], params["height"]
gen = torch.Generator().manual_seed(0)
nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_actor(W, H, gen=gen) for _ in range(4)], critics=[random_critic(W, H, gen) for _ in range(4)])
rows = 2 * n; seg = rows // 4
segs = [(i * seg, seg, i, TILE_ACT) for i in range(4)] + [(i * seg, seg, 4 + i, 0) for i in range(4)]
pp = PolicyPlan(nets, rows, segs)
obs = (torch.rand((rows, W, H, 20), device="cuda") < 0.1).to(torch.int8)
acts = torch.zeros(rows, dtype=torch.int32, device="cuda"); lp = torch.zeros(rows, device="cuda"); val = torch.zeros(rows, device="cuda")
for _ in range(3): pp.forward(obs, raw=val, actions=acts, action_logp=lp)
torch.cuda.synchronize()
L = native.lib()
L.oc_policy_debug_trace.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
assert L.oc_policy_debug_trace(1, None, 0) == 0
pp.forward(obs, raw=val, actions=acts, action_logp=lp)
nw = 256 * 8
buf = np.zeros((nw, 32), dtype=np.uint64)
assert L.oc_policy_debug_trace(0, buf.ctypes.data_as(ctypes.c_void_p), nw) == 0
for wg in (0, 100):
    w = buf[wg * 8:(wg + 1) * 8]
    t0 = w[:, 0].min()
    for i in range(8):
        hw = int(w[i, 31])
        t = (w[i, :27].astype(np.float64) - float(t0)) / 100.0
        print(f"wg {wg} wave {i} role {'C' if i < 4 else 'D'} hw_id {hw:#06x} simd {(hw >> 4) & 3} wave_slot {hw & 15} cu {(hw >> 8) & 15}: " + " ".join(f"{x:6.2f}" for x in t[:8]) + " ... " + " ".join(f"{x:6.2f}" for x in t[20:27]))
d = np.diff(buf[:, 1:22].astype(np.float64), axis=1) / 100.0
print("median us between consecutive full barriers:", np.round(np.median(d, axis=0), 2))
